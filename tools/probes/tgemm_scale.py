"""How the training GEMM's time scales with M (rows of W = output channels) and K at the fp0 shape: python tools/probes/tgemm_scale.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from patchaugnet_amd import train_ops
B, N = 18, 4096
for M, K in [(64, 256), (128, 256), (256, 256), (512, 256), (256, 64), (256, 128), (256, 512), (64, 64)]:
    W = torch.randn(M, K, device="cuda"); X = torch.randn(B, K, N, device="cuda"); Y = torch.empty(B, M, N, device="cuda")
    pblk = torch.rand(7, K, device="cuda")
    stats = torch.zeros(train_ops.STAT_SLOTS, 2, M, dtype=torch.float64, device="cuda")
    for name, fn in (("bn-loader+stats", lambda: train_ops.tgemm_nn(B, M, N, K, W, 0, K, True, X, K * N, N, Y, M * N, N, bmode=1, bp=pblk, stats=stats)),
                     ("plain", lambda: train_ops.tgemm_nn(B, M, N, K, W, 0, K, True, X, K * N, N, Y, M * N, N))):
        for _ in range(3): fn()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20): fn()
        e.record(); e.synchronize()
        ms = s.elapsed_time(e) / 20
        print(f"M={M:4d} K={K:4d} {name:16s}: {ms * 1e3:7.1f} us  {2.0 * B * M * N * K / ms / 1e9:6.1f} TFLOP/s   B-operand stream {B * K * N * 4 / ms / 1e6:7.1f} GB/s x {(M + 63) // 64} m-tiles")
