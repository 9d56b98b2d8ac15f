#!/usr/bin/env python
"""Cycle stamps of the pruned kNN kernel at the sa0 shape (b=32, n=4096, m=1024, k=20): prologue vs queries."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from patchaugnet_amd import _lib
from patchaugnet_amd.weights import synthetic_submaps
lib = _lib.lib()
lib.pa_knn_debug_buffer.argtypes = [ctypes.c_void_p]; lib.pa_knn_debug_buffer.restype = None
for kind in ("uniform", "street"):
    x = synthetic_submaps(32, 4096, 5, kind).squeeze(1).cuda().contiguous()
    q = x[:, :1024].contiguous()
    idx = torch.empty(32, 1024, 20, dtype=torch.int32, device="cuda"); d2 = torch.empty(32, 1024, 20, device="cuda")
    buf = torch.zeros(8, dtype=torch.int64, device="cuda")
    for it in range(3):
        lib.pa_knn_debug_buffer(ctypes.c_void_p(buf.data_ptr()))
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        _lib.call("pa_knnquery", 32, 4096, 1024, 20, _lib.ptr(x), _lib.ptr(q), _lib.ptr(idx), _lib.ptr(d2))
        e.record(); e.synchronize()
        lib.pa_knn_debug_buffer(None)
    t = buf.cpu().tolist()
    nq = max(t[5], 1)
    print(f"{kind}: kernel {s.elapsed_time(e)*1000:.0f} us | wave0 of block0: prologue {t[0]} cyc, queries {t[1]} cyc for {nq} queries = {t[1]//nq}/query, "
          f"chunks visited {t[2]/nq:.1f}/query, insertions {t[3]/nq:.1f}/query, sort {t[4]//nq} cyc/query")
