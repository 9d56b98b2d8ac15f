#!/usr/bin/env python
"""How long does the HOST need to issue one step (python + ctypes + allocator), vs how long the GPU needs to run it?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from patchaugnet_amd import configs, patch_aug_net
from patchaugnet_amd.extract import StreamPipeline
from patchaugnet_amd.weights import seeded_state_dict, synthetic_submaps
m = patch_aug_net.Network(param=configs.patch_aug_net_config(), use_a2a_recon=True, use_l2_norm=True)
m.load_state_dict(seeded_state_dict(m.state_dict())); m = m.cuda().eval()
m.mlp_dtype = os.environ.get("DT", "f32")
x = synthetic_submaps(32, 4096, seed=1).cuda()
with torch.no_grad():
    for s in (1, 3):
        pipe = StreamPipeline(s)
        pipe.begin()
        for _ in range(6): pipe.submit(lambda: m(x, return_feat=False))
        pipe.end(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        pipe.begin()
        for _ in range(60): pipe.submit(lambda: m(x, return_feat=False))
        t_issue = time.perf_counter() - t0
        pipe.end(); torch.cuda.synchronize()
        t_all = time.perf_counter() - t0
        print(f"streams={s}: host issue {t_issue/60*1e3:.3f} ms/step, wall {t_all/60*1e3:.3f} ms/step")
