"""Phase stamps of the persistent first-level kernel (sa_tiny.hip), split by wave group: python tools/probes/sa_tiny_phases.py"""
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
t = buf.view(512, 8).cpu().numpy()
d = t[:, 1:5] - t[:, 0:4]
for name, sel in (("waves 0-3", (np.arange(512) % 8) < 4), ("waves 4-7", (np.arange(512) % 8) >= 4)):
    print(name, "prologue/L0/L1/L2 medians:", [int(np.median(d[sel, i])) for i in range(4)], "tile total", int(np.median(t[sel, 4] - t[sel, 0])))
