#!/usr/bin/env python
"""kNN at the sa0 shape (b = 32, n = 4096, m = 1024, k = 20) on uniform and street-like clouds: kernel time with HIP events.
   A/B: PA_KNN_LANE=1 python tools/knn_time.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from patchaugnet_amd import _lib
from patchaugnet_amd.weights import synthetic_submaps
for kind in ("uniform", "street"):
    x = synthetic_submaps(32, 4096, 5, kind).squeeze(1).cuda().contiguous()
    idx0 = torch.empty(32, 1024, dtype=torch.int32, device="cuda"); q = torch.empty(32, 1024, 3, device="cuda")
    _lib.call("pa_furthestsampling_gather", 32, 4096, 1024, _lib.ptr(x), _lib.ptr(idx0), _lib.ptr(q))
    idx = torch.empty(32, 1024, 20, dtype=torch.int32, device="cuda"); d2 = torch.empty(32, 1024, 20, device="cuda")
    fn = lambda: _lib.call("pa_knnquery", 32, 4096, 1024, 20, _lib.ptr(x), _lib.ptr(q), _lib.ptr(idx), _lib.ptr(d2))
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): fn()
    e.record(); e.synchronize()
    print(f"{kind}: pa_knnquery(32, 4096, 1024, 20) {s.elapsed_time(e) / 20 * 1000:.1f} us  (lane {'on' if os.environ.get('PA_KNN_LANE') else 'off'}, quad {'off' if os.environ.get('PA_KNN_NO_QUAD') else 'on'})")
    import ctypes
    lib = _lib.lib(); lib.pa_knn_debug_buffer.argtypes = [ctypes.c_void_p]; lib.pa_knn_debug_buffer.restype = None
    buf = torch.zeros(16, dtype=torch.int64, device="cuda")
    lib.pa_knn_debug_buffer(ctypes.c_void_p(buf.data_ptr())); fn(); torch.cuda.synchronize(); lib.pa_knn_debug_buffer(None)
    t = buf.cpu().tolist()
    if not os.environ.get("PA_KNN_LANE") and not os.environ.get("PA_KNN_NO_QUAD"):
        print(f"   quad kernel, block 0 thread 0: cloud sort {t[1]-t[0]} cyc, query sort {t[2]-t[1]}, pass1 {t[3]-t[2]}, pass2 {t[4]-t[3]}, pass3 {t[5]-t[4]}; R = {t[8]}, keys {t[9]}, overflow {t[10]}")
    if os.environ.get("PA_KNN_LANE"):
        print(f"   block 0, thread 0: sort {t[1]-t[0]} cyc, pass1 {t[2]-t[1]}, pass2 {t[3]-t[2]}, pass3 {t[4]-t[3]}, store {t[5]-t[4]}; R = {t[6]}, queued {t[7]}; pass-1 candidates per lane: max {t[8]}, mean {t[9] / 64:.0f}")
    if _lib.has("pa_kg_stamps_read"):      # variant build with -DKG_STAMPS (tools/build_variant.sh): phases of the cloud sort
        h = (ctypes.c_longlong * 16)(); lib.pa_kg_stamps_read.argtypes = [ctypes.c_void_p]; lib.pa_kg_stamps_read(h)
        t = list(h)
        print("   cloud sort phases (cycles): load+box %d, reduce %d, cells+histogram %d, scan %d, scatter %d, chunk boxes %d" % tuple(t[i + 1] - t[i] for i in range(6)))
