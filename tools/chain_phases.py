#!/usr/bin/env python
"""Per-phase cycle stamps of the fp0 chain kernel (profiling aid): prologue / layer0 / layer1 / layer2+epilogue."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from patchaugnet_amd import _lib, configs, patch_aug_net
from patchaugnet_amd.weights import seeded_state_dict, synthetic_submaps

model = patch_aug_net.Network(param=configs.patch_aug_net_config(), use_a2a_recon=True, use_l2_norm=True)
model.load_state_dict(seeded_state_dict(model.state_dict()))
model = model.cuda().eval()
x = synthetic_submaps(32, 4096, seed=1234).cuda()
lib = _lib.lib()
lib.pa_chain_debug_buffer.argtypes = [ctypes.c_void_p]
lib.pa_chain_debug_buffer.restype = None
with torch.no_grad():
    for _ in range(2):
        model(x, return_feat=False)
    eng = model._engine
    name = "fp_premul" if eng.premul else "fp"
    orig = getattr(eng.fp[0], name)
    buf = torch.zeros(512 * 8, dtype=torch.int64, device="cuda")

    def wrapped(*a, **k):
        lib.pa_chain_debug_buffer(ctypes.c_void_p(buf.data_ptr()))
        r = orig(*a, **k)
        lib.pa_chain_debug_buffer(None)
        return r
    setattr(eng.fp[0], name, wrapped)
    model(x, return_feat=False)
    torch.cuda.synchronize()
t = buf.view(512, 8).cpu().numpy()
nl = 3 if eng.premul else 4
d = t[:, 1:nl + 1] - t[:, 0:nl]
import numpy as np
names = ["prologue", "layer0", "layer1", "layer2+epilogue"] if nl == 4 else ["prologue(+folded layer)", "layer1", "layer2+epilogue"]
for i, n in enumerate(names):
    print(f"{n:18s} median {np.median(d[:, i]):10.0f}  min {d[:, i].min():10.0f}  max {d[:, i].max():10.0f}  (counter ticks)")
print("total median", np.median(t[:, nl] - t[:, 0]), " start spread", t[:, 0].max() - t[:, 0].min())
