#!/usr/bin/env python
"""NetVLAD at the finest scale (b = 32, n = 4096, c = 256, k = 64): pa_netvlad_rows launch time with HIP events, checked against fp64 torch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from patchaugnet_amd import _lib, configs, patch_aug_net
from patchaugnet_amd.engine import _Vlad
from patchaugnet_amd.weights import seeded_state_dict

m = patch_aug_net.Network(param=configs.patch_aug_net_config(), use_a2a_recon=True, use_l2_norm=True)
m.load_state_dict(seeded_state_dict(m.state_dict()))
m = m.cuda().eval()
vl = [mod for mod in m.modules() if hasattr(mod, "cluster_weights")]
for v in vl:
    ev = _Vlad(v, torch.device("cuda"))
    b, n, k = 32, ev.n, ev.k
    x = torch.randn(b, n, 256, device="cuda")
    out = torch.zeros(b, k, 256, device="cuda")
    fn = lambda: ev.run(x, out, k, 0, rows=True)
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = []
    for _ in range(5):
        s.record()
        for _ in range(20): fn()
        e.record(); e.synchronize()
        reps.append(s.elapsed_time(e) / 20 * 1000)
    with torch.no_grad():
        want = v(x.transpose(1, 2).unsqueeze(-1).contiguous())          # (B, C, K)?
    got = out.transpose(1, 2) if want.shape == (b, 256, k) else out
    err = (got.reshape(want.shape) - want).abs().max().item() if got.numel() == want.numel() else float("nan")
    print(f"netvlad n={n} k={k}: {min(reps):.1f} us per call (accumulate + finalize; min of {[round(r, 1) for r in reps]}), max |err| vs module {err:.2e}")
import ctypes
lib = _lib.lib()
if hasattr(lib, "pa_vlad_debug_read"):
    buf = (ctypes.c_longlong * 16)()
    torch.cuda.synchronize(); lib.pa_vlad_debug_read(buf)
    t = list(buf)
    names = ["lds store", "barrier", "fetch issue", "assign gemm", "softmax", "barrier", "aggregate gemm", "barrier"]
    print("   second tile of block 0, thread 0 (cycles): " + ", ".join(f"{n} {t[i+1]-t[i]}" for i, n in enumerate(names)) + f"; tile total {t[8]-t[0]}")
    print(f"   whole kernel {t[12]-t[9]} cycles: prologue to loop {t[10]-t[9]}, loop {t[11]-t[10]}, epilogue {t[12]-t[11]}")
