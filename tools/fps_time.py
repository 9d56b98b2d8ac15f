#!/usr/bin/env python
"""FPS at the three levels of a batch (b = 32): kernel time with HIP events + bit-exactness of the variants against the default build's output file."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from patchaugnet_amd import _lib
from patchaugnet_amd.weights import synthetic_submaps
for kind in ("uniform", "street"):
    x = synthetic_submaps(32, 4096, 5, kind).squeeze(1).cuda().contiguous()
    for n, m in ((4096, 1024), (1024, 128)):
        xs = x[:, :n].contiguous()
        idx = torch.empty(32, m, dtype=torch.int32, device="cuda"); q = torch.empty(32, m, 3, device="cuda")
        fn = lambda: _lib.call("pa_furthestsampling_gather", 32, n, m, _lib.ptr(xs), _lib.ptr(idx), _lib.ptr(q))
        for _ in range(3): fn()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10): fn()
        e.record(); e.synchronize()
        print(f"{kind} n={n} m={m}: {s.elapsed_time(e) / 10 * 1000:.1f} us  ({s.elapsed_time(e) / 10 * 1000 / m:.3f} us/round)  checksum {int(idx.long().sum())} {int((idx.long() * torch.arange(m, device='cuda')).sum())}")
