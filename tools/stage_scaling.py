#!/usr/bin/env python
"""Stage times of the fused engine at several batch sizes (rows per launch vs achieved MFMA rate): python tools/stage_scaling.py [B ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from patchaugnet_amd import configs, patch_aug_net, profiling
from patchaugnet_amd.weights import seeded_state_dict, synthetic_submaps

model = patch_aug_net.Network(param=configs.patch_aug_net_config(), use_a2a_recon=True, use_l2_norm=True)
model.load_state_dict(seeded_state_dict(model.state_dict()))
model = model.cuda().eval()
GF32 = {"sa0.chain": 4.28, "sa1.chain": 4.06, "sa2.chain": 5.38, "fp2.chain": 2.15, "fp1.premul": 0.54, "fp1.chain": 5.37, "fp0.premul": 4.29, "fp0.chain": 34.56, "vlad": 9.14}
for b in [int(v) for v in sys.argv[1:]] or [32, 64, 128, 256]:
    x = synthetic_submaps(b, 4096, seed=1234).cuda()
    with torch.no_grad():
        st = profiling.stage_times(model, x, 5)
    print(f"B={b}: " + "  ".join(f"{k}={v * 1e3:.1f}us" + (f"({GF32[k] * b / 32 / v / 157.3:.2f})" if k in GF32 else "") for k, v in st.items()))
    del x
    torch.cuda.empty_cache()
