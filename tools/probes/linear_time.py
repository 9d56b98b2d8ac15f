"""Time pa_linear (k = 256) with and without the LDS-resident-weights kernel: python tools/probes/linear_time.py [rows n]..."""
import ctypes, sys
import torch
from patchaugnet_amd import _lib
from patchaugnet_amd._lib import call, ptr
from patchaugnet_amd.engine import pack_weights

lib = _lib.lib()
lib.pa_linear_lds_enable.argtypes, lib.pa_linear_lds_enable.restype = [ctypes.c_int], None
shapes = [(4096, 256), (32768, 256), (32768, 512), (131072, 256)]
for rows, n in shapes:
    x = torch.randn(rows, 256, device="cuda"); wt = torch.randn(256, n, device="cuda") * 0.06; b = torch.zeros(n, device="cuda")
    out = torch.empty(rows, n, device="cuda")
    wp = pack_weights(wt)
    for on in (0, 1):
        lib.pa_linear_lds_enable(on)
        for _ in range(5):
            call("pa_linear", rows, 256, n, ptr(x), 256, ptr(wt), ptr(wp), ptr(b), 0, None, 0, ptr(out), n)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            call("pa_linear", rows, 256, n, ptr(x), 256, ptr(wt), ptr(wp), ptr(b), 0, None, 0, ptr(out), n)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 20
        print(f"rows {rows} n {n} lds={on}: {us:.1f} us  {2*rows*256*n/us/1e6:.1f} TFLOP/s")
lib.pa_linear_lds_enable(-1)
