#!/usr/bin/env python
"""Random shapes through pa_tgemm_nn / pa_tgemm_nn_bnred on the LDS-resident-weights kernel (forced) against the LDS-tiled kernel: every operand
transform, both layouts of A, bias, beta, statistics, fused BatchNorm-backward sums, ragged row blocks, both column-tile widths, repeated launches.
python tools/fuzz_tgemm_cm.py [cases] [seed]"""
import ctypes, os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from patchaugnet_amd import _lib, train_ops as T
from patchaugnet_amd._lib import call, ptr
lib = _lib.lib()
lib.pa_tgemm_cm_enable.argtypes, lib.pa_tgemm_cm_enable.restype = [ctypes.c_int], None
cases, seed = int(sys.argv[1]) if len(sys.argv) > 1 else 150, int(sys.argv[2]) if len(sys.argv) > 2 else 1
rnd = random.Random(seed)
bad = 0
for case in range(cases):
    K = rnd.choice([32, 64, 128, 256])
    M = rnd.choice([32, 64, 96, 128, 160, 192, 256, 320, 512])
    N = 32 * rnd.randint(1, 96) * rnd.choice([1, 1, 2, 8])
    B = rnd.randint(1, 6) if N > 4096 else rnd.randint(1, 20)
    if B * M * N * 4 > 1.5e9 or B * K * N * 4 > 1.5e9:
        continue
    mode = rnd.choice([0, 1, 2, 3])
    kc = rnd.choice([True, False])
    kind = rnd.choice(["plain", "stats", "bias", "beta", "bnred"])
    if kind == "stats" and mode >= 2: kind = "plain"
    if kind == "bnred" and mode < 2: kind = "plain"
    g = torch.Generator().manual_seed(case + 1000 * seed)
    A = ((torch.randn(M, K, generator=g) if kc else torch.randn(K, M, generator=g)) / K ** 0.5).cuda()
    X = torch.randn(B, K, N, generator=g).cuda()
    aux = torch.randn(B, K, N, generator=g).cuda() if mode >= 2 else None
    p = torch.randn(7, K, generator=g); p[3] = p[3].abs() + 0.5; p = p.cuda().contiguous()
    bias = torch.randn(M, generator=g).cuda() if kind == "bias" else None
    C0 = torch.randn(B, M, N, generator=g).cuda()
    ynext = torch.randn(B, M, N, generator=g).cuda() if kind == "bnred" else None
    pnext = torch.randn(7, M, generator=g); pnext[3] = pnext[3].abs() + 0.5; pnext = pnext.cuda().contiguous()
    res = []
    for on in (0, 1, 1):
        lib.pa_tgemm_cm_enable(on)
        C = C0.clone()
        st = torch.zeros(T.STAT_SLOTS, 2, M, dtype=torch.float64, device="cuda") if kind == "stats" else None
        sums = torch.zeros(2 * M, dtype=torch.float64, device="cuda") if kind == "bnred" else None
        if kind == "bnred":
            if on == 0: relu = rnd.choice([0, 1])
            call("pa_tgemm_nn_bnred", B, M, N, K, ptr(A), K if kc else M, int(kc), ptr(X), K * N, N, mode, ptr(aux), ptr(p), ptr(C), M * N, N, ptr(ynext),
                 ptr(pnext), relu, ptr(sums))
        else:
            T.tgemm_nn(B, M, N, K, A, 0, K if kc else M, kc, X, K * N, N, C, M * N, N, bmode=mode, baux=aux, bp=p if mode else None, bias=bias,
                       beta=1 if kind == "beta" else 0, stats=st)
        torch.cuda.synchronize()
        res.append((C, st.sum(0) if st is not None else None, sums))
    lib.pa_tgemm_cm_enable(-1)
    scale = max(res[0][0].abs().max().item(), 1.0)
    ok = True
    for r in res[1:]:
        d = (r[0] - res[0][0]).abs().max().item()
        if not d <= 3e-5 * scale: ok = False
        if r[1] is not None and not torch.allclose(r[1], res[0][1], rtol=1e-5, atol=1e-4 * scale * (B * N) ** 0.5): ok = False
        if r[2] is not None and not torch.allclose(r[2], res[0][2], rtol=1e-5, atol=1e-4 * scale * (B * N) ** 0.5): ok = False
    if not ok:
        bad += 1
        print(f"MISMATCH case {case}: B={B} M={M} N={N} K={K} mode={mode} kc={kc} kind={kind} max diff {(res[1][0] - res[0][0]).abs().max().item():.3e} / {(res[2][0] - res[0][0]).abs().max().item():.3e} scale {scale:.2f}")
print(f"{cases} cases, {bad} mismatches")
