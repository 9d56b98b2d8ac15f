"""Per-wave stamps of the persistent first-level kernel (sa_tiny.hip, register form): python tools/probes/sa_tiny_phases.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
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
    chain = model._engine.sa[0]
    orig = chain.sa
    buf = torch.zeros(512 * 8, dtype=torch.int64, device="cuda")
    def wrapped(*a, **k):
        lib.pa_chain_debug_buffer(ctypes.c_void_p(buf.data_ptr()))
        r = orig(*a, **k)
        lib.pa_chain_debug_buffer(None)
        return r
    chain.sa = wrapped
    model(x, return_feat=False)
    torch.cuda.synchronize()
t = buf.view(512, 8).cpu().numpy().astype(np.int64)
t0 = t[:, 0].min()
names = ["entry", "weights in place", "first tile gathered", "tile 1 done", "tile 2 done", "tile 3 done", "tile 4 done"]
for grp, sel in (("waves 0-3 of a workgroup", (np.arange(512) % 8) < 4), ("waves 4-7 (staggered)", (np.arange(512) % 8) >= 4)):
    print(grp)
    for i, nm in enumerate(names):
        v = t[sel, i] - t0
        print(f"   {nm:22s} median {int(np.median(v)):8d}  min {int(v.min()):8d}  max {int(v.max()):8d}   (cycles after the first wave's entry; 100 MHz ticks if < 10 k total)")
