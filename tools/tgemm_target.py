#!/usr/bin/env python
"""The training GEMM at the finest FP level's 256 -> 256 forward shape (18 clouds x 4096 points), a few launches: target for rocprofv3 --pmc
and for HIP-event timing.  python tools/tgemm_target.py [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from patchaugnet_amd import train_ops
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 6
B, M, N, K = int(sys.argv[2]) if len(sys.argv) > 2 else 18, 256, 4096, 256
W = torch.randn(M, K, device="cuda"); X = torch.randn(B, K, N, device="cuda"); Y = torch.empty(B, M, N, device="cuda")
pblk = torch.rand(7, K, device="cuda")
stats = torch.zeros(train_ops.STAT_SLOTS, 2, M, dtype=torch.float64, device="cuda")
fn = lambda: train_ops.tgemm_nn(B, M, N, K, W, 0, K, True, X, K * N, N, Y, M * N, N, bmode=1, bp=pblk, stats=stats)
for _ in range(3): fn()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(iters): fn()
e.record(); e.synchronize()
ms = s.elapsed_time(e) / iters
print(f"tgemm_nn {B} x (256 x 4096 x 256), BatchNorm + ReLU loader, statistics epilogue: {ms * 1e3:.1f} us, {2.0 * B * M * N * K / ms / 1e9:.1f} TFLOP/s")
