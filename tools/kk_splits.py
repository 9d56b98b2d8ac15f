#!/usr/bin/env python
"""tgemm_kk at the skinny weight-gradient shapes of the first set-abstraction level for a forced number of k splits (PA_KK_SPLITS)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from patchaugnet_amd import train_ops as T
for (b, M, N, K) in ((18, 32, 6, 20480), (18, 32, 32, 20480), (18, 64, 32, 20480), (18, 256, 256, 4096), (18, 256, 64, 2560)):
    A = torch.randn(b, M, K, device="cuda"); aux = torch.randn(b, M, K, device="cuda"); B = torch.randn(b, N, K, device="cuda")
    pa = torch.rand(7, M, device="cuda"); pb = torch.rand(7, N, device="cuda"); C = torch.zeros(M, N, device="cuda")
    fn = lambda: T.tgemm_kk(b, M, N, K, A, M * K, K, B, N * K, K, C, 0, N, amode=2, aaux=aux, ap=pa, bmode=1, bp=pb)
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): fn()
    e.record(); e.synchronize()
    print(f"splits={os.environ.get('PA_KK_SPLITS','auto'):>5s} ({b},{M},{N},{K}): {s.elapsed_time(e)/20*1000:7.1f} us")
