#!/usr/bin/env python
"""pa_tgemm_nn at the training step's chip-filling shapes: the LDS-tiled kernel against the LDS-resident-weights kernel (csrc/train_gemm_cm.hip),
HIP events around 10 launches each, alternating.  python tools/tgemm_cm_time.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from patchaugnet_amd import _lib, train_ops as T
lib = _lib.lib()
lib.pa_tgemm_cm_enable.argtypes, lib.pa_tgemm_cm_enable.restype = [ctypes.c_int], None
shapes = [("fp0 fwd", 18, 256, 4096, 256, True, 1, True), ("fp0 dX", 18, 256, 4096, 256, False, 2, False), ("fp1 fwd", 18, 256, 1024, 256, True, 1, True),
          ("fp1 dX", 18, 256, 1024, 256, False, 2, False), ("vlad assign", 18, 64, 4096, 256, False, 0, True), ("vlad dX", 18, 256, 4096, 64, True, 3, False),
          ("sa1 L3 fwd", 18, 256, 2560, 64, True, 1, True), ("sa1 dX", 18, 64, 2560, 256, False, 2, False), ("sa2 L3", 18, 512, 320, 256, True, 1, True),
          ("fp2 L2", 18, 256, 128, 256, True, 1, True), ("sa0 L3 fwd", 18, 64, 20480, 32, True, 1, True), ("sa0 L2 fwd", 18, 32, 20480, 32, True, 1, True),
          ("sa0 L3 dX", 18, 32, 20480, 64, False, 2, False), ("sa0 L2 dX", 18, 32, 20480, 32, False, 2, False)]
if os.environ.get('PA_TGEMM_CM_DECOMP'):          # the fp0 shape without the operand transform and / or the statistics epilogue
    shapes = [("fp0 fwd", 18, 256, 4096, 256, True, 1, True), ("fp0 tf only", 18, 256, 4096, 256, True, 1, False), ("fp0 stats only", 18, 256, 4096, 256, True, 0, True),
              ("fp0 plain", 18, 256, 4096, 256, True, 0, False), ("fp0 plain 36", 36, 256, 4096, 256, True, 0, False)]
if os.environ.get('PA_TGEMM_CM_SWEEP'):          # time against batch: intercept = what a launch costs beyond its tiles
    shapes = [(f"fp0 plain {b}", b, 256, 4096, 256, True, 0, False) for b in (6, 12, 18, 24, 36, 54)] + [(f"fp0 fwd {b}", b, 256, 4096, 256, True, 1, True) for b in (6, 12, 18, 24, 36, 54)]
only = os.environ.get('PA_TGEMM_CM_ONLY')
for name, B, M, N, K, kc, bmode, stats in shapes:
    if only and not name.startswith(only):
        continue
    g = torch.Generator().manual_seed(1)
    A = (torch.randn(M, K, generator=g) if kc else torch.randn(K, M, generator=g)).cuda() / K ** 0.5
    X = torch.randn(B, K, N, generator=g).cuda()
    aux = torch.randn(B, K, N, generator=g).cuda() if bmode >= 2 else None
    p = (torch.rand(7, K, generator=g) + 0.25).cuda() if bmode else None
    C = torch.empty(B, M, N, device="cuda")
    st = torch.zeros(T.STAT_SLOTS, 2, M, dtype=torch.float64, device="cuda") if stats else None
    fn = lambda: T.tgemm_nn(B, M, N, K, A, 0, K if kc else M, kc, X, K * N, N, C, M * N, N, bmode=bmode, baux=aux, bp=p, stats=st)
    res = {}
    for rep in range(2):
        for on in (0, 1):
            lib.pa_tgemm_cm_enable(on)
            for _ in range(3): fn()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(10): fn()
            e.record(); e.synchronize()
            res.setdefault(on, []).append(s.elapsed_time(e) / 10 * 1e3)
    lib.pa_tgemm_cm_enable(-1)
    fl = 2.0 * B * M * N * K
    t0, t1 = min(res[0]), min(res[1])
    print(f"{name:12s} B={B} M={M} N={N} K={K} mode={bmode}{' stats' if stats else ''}: lds-tiled {t0:7.1f} us ({fl / t0 / 1e6:5.1f} TF)   lds-resident {t1:7.1f} us ({fl / t1 / 1e6:5.1f} TF = {fl / t1 / 1e6 / 157.3:.2f} of peak)")
