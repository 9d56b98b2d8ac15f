#!/usr/bin/env python
"""3-NN + weights at the finest FP level (b = 32, 4096 unknown, 1024 known): grid kernel vs brute-force scan.  PA_TNN_NO_GRID=1 for the scan."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from patchaugnet_amd import _lib
from patchaugnet_amd.weights import synthetic_submaps
for kind in ("uniform", "street"):
    x = synthetic_submaps(32, 4096, 5, kind).squeeze(1).cuda().contiguous()
    for n, m in ((4096, 1024), (1024, 128)):
        xs = x[:, :n].contiguous()
        idx0 = torch.empty(32, m, dtype=torch.int32, device="cuda"); kn = torch.empty(32, m, 3, device="cuda")
        _lib.call("pa_furthestsampling_gather", 32, n, m, _lib.ptr(xs), _lib.ptr(idx0), _lib.ptr(kn))
        w = torch.empty(32, n, 3, device="cuda"); idx = torch.empty(32, n, 3, dtype=torch.int32, device="cuda")
        fn = lambda: _lib.call("pa_three_nn_weights", 32, n, m, _lib.ptr(xs), _lib.ptr(kn), _lib.ptr(w), _lib.ptr(idx))
        for _ in range(3): fn()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20): fn()
        e.record(); e.synchronize()
        print(f"{kind} n={n} m={m}: pa_three_nn_weights {s.elapsed_time(e) / 20 * 1000:.1f} us  (grid {'off' if os.environ.get('PA_TNN_NO_GRID') else 'on'})")
