#!/usr/bin/env python
"""Randomised shape fuzzing of the HIP ops against the CPU oracle / fp64 torch.  A bounded slice (a few seconds per family, fixed seed) is
collected by `pytest -m gpu` through tests/test_gpu_fuzz.py; longer campaigns are run by hand on the GPU box:
    python tests/fuzz_gpu.py [seconds per op family, default 20]
Index outputs and pure gathers must be bit-exact; GEMM-based ops within 2e-5 relative of an fp64 reference."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from patchaugnet_amd.hostcpu import limit_host_threads
limit_host_threads()
from oracle import oracle_ops as o
from patchaugnet_amd import pointops as P, _lib
from patchaugnet_amd._lib import call, ptr
from patchaugnet_amd.engine import pack_weights

BUDGET = 20.0            # seconds per family; the command line / tests/test_gpu_fuzz.py override it
rng = np.random.default_rng(int(os.environ.get("FUZZ_SEED", "2026")))


def reseed(seed):
    global rng
    rng = np.random.default_rng(seed)


def cloud(b, n):
    kind = rng.integers(0, 4)
    x = (rng.random((b, n, 3), dtype=np.float32) * 2 - 1).astype(np.float32)
    if kind == 1:
        x = (np.round(x * rng.integers(2, 9)) / 8).astype(np.float32)
    elif kind == 2 and n > 4:
        k = max(n // 8, 1)
        x[:, rng.choice(n, k, replace=False)] = x[:, rng.choice(n, k, replace=False)]
    elif kind == 3:
        x[..., 2] = 0.0          # planar
    return x


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def logint(lo, hi):
    return int(np.exp(rng.uniform(np.log(lo), np.log(hi + 1))))


def run(name, fn, budget=None):
    t0, cases = time.time(), 0
    while time.time() - t0 < (BUDGET if budget is None else budget):
        desc = fn()
        cases += 1
        if desc is not None:
            print(f"FAIL {name}: {desc}", flush=True)
            return False
    print(f"ok   {name}: {cases} random cases", flush=True)
    return True


def f_fps():
    b, n = int(rng.integers(1, 4)), logint(1, 6000)
    m = int(rng.integers(1, min(n, 1200) + 1))
    x = cloud(b, n)
    if not np.array_equal(P.furthestsampling(dev(x), m).cpu().numpy(), o.furthestsampling(x, m)):
        return f"b={b} n={n} m={m}"


def f_knn():
    b, n, m = int(rng.integers(1, 4)), logint(1, 5000), logint(1, 1100)
    k = int(rng.integers(1, 65))
    x, q = cloud(b, n), cloud(b, m)
    if rng.random() < 0.5:
        mm = min(m, n)
        q[:, :mm] = x[:, :mm]
    ref = o.knnquery(k, x, q)
    got = P.knnquery(k, dev(x), dev(q)).cpu().numpy()
    if not np.array_equal(got, ref[0] if isinstance(ref, tuple) else ref):
        return f"b={b} n={n} m={m} k={k}"


def hard_cloud(b, n):
    """clouds that stress the cell-grid kernels: everything cloud() draws, plus dense clusters with far outliers, anisotropic boxes,
    tiny scales, large offsets (coarse fp32 grid of coordinates), a line"""
    kind = rng.integers(0, 9)
    if kind < 4:
        return cloud(b, n)
    x = (rng.random((b, n, 3), dtype=np.float32) * 2 - 1).astype(np.float32)
    if kind == 4 and n > 50:
        x[:, : n - n // 50] = (x[:, : n - n // 50] * rng.choice([0.01, 0.05, 0.2]) + rng.uniform(-0.8, 0.8, 3)).astype(np.float32)
    elif kind == 5:
        x = (x * np.array([1.0, rng.choice([0.01, 0.1]), rng.choice([1e-3, 0.5])], dtype=np.float32)).astype(np.float32)
    elif kind == 6:
        x = (x * np.float32(rng.choice([1e-3, 1e-2, 30.0]))).astype(np.float32)
    elif kind == 7:
        x = (x + np.float32(rng.choice([100.0, 1000.0, -5000.0]))).astype(np.float32)
    elif kind == 8:
        x[..., 1] = x[..., 0] * 0.5
        x[..., 2] = -0.25
    return x


def f_knn_grid():
    """the four-lanes-per-query kernel's domain (knn_quad.hip): 2048..4096 source points, >= 256 queries, 16 / 20 / 32 neighbours"""
    b, n, m = int(rng.integers(1, 3)), int(rng.integers(2048, 4097)), int(rng.integers(256, 1300))
    k = int(rng.choice([16, 20, 32]))
    x = hard_cloud(b, n)
    mode = rng.integers(0, 3)
    if mode == 0:
        q = x[:, rng.choice(n, min(m, n), replace=False)].copy()
    elif mode == 1:
        q = hard_cloud(b, m)
    else:                                   # queries around the cloud, some far outside its box
        q = (x[:, rng.choice(n, m)] + rng.standard_normal((b, m, 3)).astype(np.float32) * np.float32(rng.choice([1e-4, 0.05, 3.0]))).astype(np.float32)
    ri, rd = o.knnquery(k, x, q)
    gi, gd = P.knnquery_with_dist(k, dev(x), dev(q))
    if not (np.array_equal(gi.cpu().numpy(), ri) and np.array_equal(gd.cpu().numpy().view(np.uint32), rd.view(np.uint32))):
        return f"b={b} n={n} m={q.shape[1]} k={k} mode={mode}"


def f_3nn_grid():
    """the cell-grid 3-NN's domain (three_nn_grid.hip): 512..4096 known points, >= 1024 queries"""
    b, m, n = int(rng.integers(1, 3)), int(rng.integers(512, 4097)), int(rng.integers(1024, 5000))
    kn = hard_cloud(b, m)
    mode = rng.integers(0, 3)
    if mode == 0:
        u = hard_cloud(b, n)
    elif mode == 1:
        u = (kn[:, rng.choice(m, n)] + rng.standard_normal((b, n, 3)).astype(np.float32) * np.float32(rng.choice([0.0, 1e-4, 0.05, 3.0]))).astype(np.float32)
    else:
        u = np.concatenate([kn, hard_cloud(b, n)], 1)[:, :n].copy()
    rd, ri = o.nearestneighbor(u, kn)
    gd, gi = P.nearestneighbor(dev(u), dev(kn))
    if not (np.array_equal(gi.cpu().numpy(), ri) and np.array_equal(gd.cpu().numpy(), np.sqrt(rd))):
        return f"b={b} n={n} m={m} mode={mode}"


def f_3nn():
    b, n, m = int(rng.integers(1, 4)), logint(1, 5000), logint(1, 3000)
    u, kn = cloud(b, n), cloud(b, m)
    rd, ri = o.nearestneighbor(u, kn)
    gd, gi = P.nearestneighbor(dev(u), dev(kn))
    if not (np.array_equal(gi.cpu().numpy(), ri) and np.array_equal(gd.cpu().numpy(), np.sqrt(rd))):
        return f"b={b} n={n} m={m}"


def f_gather():
    b, c, n, m, k = int(rng.integers(1, 4)), logint(1, 300), logint(1, 20000), logint(1, 600), int(rng.integers(1, 33))
    if b * c * m * k > 3e7:
        return None
    f = rng.standard_normal((b, c, n)).astype(np.float32)
    idx = rng.integers(0, n, (b, m, k), dtype=np.int32)
    if not np.array_equal(P.grouping(dev(f), dev(idx)).cpu().numpy(), o.grouping_forward(f, idx)):
        return f"grouping b={b} c={c} n={n} m={m} k={k}"
    i1 = rng.integers(0, n, (b, m), dtype=np.int32)
    if not np.array_equal(P.gathering(dev(f), dev(i1)).cpu().numpy(), o.gathering_forward(f, i1)):
        return f"gathering b={b} c={c} n={n} m={m}"
    i3 = rng.integers(0, n, (b, m, 3), dtype=np.int32)
    w = rng.random((b, m, 3), dtype=np.float32)
    if not np.array_equal(P.interpolation(dev(f), dev(i3), dev(w)).cpu().numpy(), o.interpolation_forward(f, i3, w)):
        return f"interpolation b={b} c={c} n={n} m={m}"


def f_backward():
    b, c, n, m, k = int(rng.integers(1, 4)), logint(1, 300), logint(1, 20000), logint(1, 600), int(rng.integers(1, 21))
    if b * c * m * k > 2e7:
        return None
    f = torch.randn(b, c, n, device="cuda", requires_grad=True)
    idx = dev(rng.integers(0, n, (b, m, k), dtype=np.int32))
    go = torch.randn(b, c, m, k, device="cuda")
    P.grouping(f, idx).backward(go)
    tol = 1e-4 + 4e-6 * (m * k / n)        # fp32 sums of ~m*k/n addends per bin in a different order than the oracle
    if not np.allclose(f.grad.cpu().numpy(), o.grouping_backward(go.cpu().numpy(), idx.cpu().numpy(), n), rtol=1e-4, atol=tol):
        return f"grouping_backward b={b} c={c} n={n} m={m} k={k}"
    f.grad = None
    i3 = dev(rng.integers(0, n, (b, m, 3), dtype=np.int32))
    w = torch.rand(b, m, 3, device="cuda")
    g3 = torch.randn(b, c, m, device="cuda")
    P.interpolation(f, i3, w).backward(g3)
    if not np.allclose(f.grad.cpu().numpy(), o.interpolation_backward(g3.cpu().numpy(), i3.cpu().numpy(), w.cpu().numpy(), n), rtol=1e-4, atol=tol):
        return f"interpolation_backward b={b} c={c} n={n} m={m}"


def f_linear():
    rows, k, n = logint(1, 5000), logint(1, 600), 16 * int(rng.integers(1, 65))
    relu, res, packed = int(rng.integers(0, 2)), bool(rng.integers(0, 2)), n % 64 == 0 and bool(rng.integers(0, 2))
    x = torch.randn(rows, k, device="cuda")
    w = torch.randn(n, k, device="cuda") / k ** 0.5
    bias = torch.randn(n, device="cuda")
    r = torch.randn(rows, n, device="cuda") if res else None
    kpad = (k + 3) // 4 * 4
    wt = torch.zeros(kpad, n, device="cuda")
    wt[:k] = w.t()
    out = torch.empty(rows, n, device="cuda")
    call("pa_linear", rows, k, n, ptr(x), k, ptr(wt), ptr(pack_weights(wt)) if packed else None, ptr(bias), relu, ptr(r), n if res else 0, ptr(out), n)
    ref = x.double() @ w.double().t() + bias.double()
    if relu:
        ref = ref.clamp_min(0)
    if res:
        ref = ref + r.double()
    err = (out.double() - ref).abs().max().item()
    if not err <= 2e-5 * max(ref.abs().max().item(), 1.0):
        return f"rows={rows} k={k} n={n} relu={relu} res={res} packed={packed} err={err}"


def f_attention():
    from patchaugnet_amd import backbone
    from patchaugnet_amd.engine import _Attn
    from oracle import models_cpu
    b, n, c = int(rng.integers(1, 4)), logint(1, 1100), int(rng.choice([64, 128, 256, 512]))
    if n * c > 300000:
        n = max(1, 300000 // c)
    torch.manual_seed(int(rng.integers(0, 1 << 30)))
    sa = backbone.SALayer(c, 8).eval()
    for p in sa.parameters():
        p.data.mul_(0.5)
    x = torch.randn(b, c, n)
    sd = {"s." + k: v for k, v in sa.state_dict().items()}
    with torch.no_grad():
        ref = models_cpu.sa_layer(sd, "s", x, 8)
        xm = x.transpose(1, 2).contiguous().view(b * n, c).cuda()
        got = _Attn(sa, xm.device).run(xm, b, n).view(b, n, c).transpose(1, 2).cpu()
    err = (got - ref).abs().max().item()
    if not err <= 5e-5 * max(ref.abs().max().item(), 1.0):
        return f"b={b} n={n} c={c} err={err}"


def _chain_dims(k0, pooled):
    """Random legal widths: hidden layers 16*2^j (single chunk), <= 64 when pooled wave-private, last layer a multiple of 16."""
    nl = int(rng.integers(1, 4))
    hid_choices = [16, 32, 64] if pooled else [16, 32, 64, 128, 256]
    dims = [k0] + [int(rng.choice(hid_choices)) for _ in range(nl - 1)] + [16 * int(rng.integers(1, 33))]
    return dims


def f_chain_sa():
    from tests.test_gpu_chain import make_layers, mlp_ref, sa_inputs, sa_rows_ref
    from patchaugnet_amd.engine import _Chain
    B, n, ns = int(rng.integers(1, 4)), logint(2, 3000), int(rng.choice([13, 16, 17, 20, 29, 32]))
    m, C = logint(1, min(n, 400)), int(rng.choice([3, 5, 8, 61, 64, 128]))
    pooled = bool(rng.integers(0, 2))
    if not pooled:
        ns = int(rng.integers(1, 33))
    if B * m * ns * C > 4e6:
        return None
    dims = _chain_dims(3 + C, pooled)
    seed = int(rng.integers(0, 1 << 30))
    ref, eng = make_layers(dims, seed=seed)
    xyz, feat, cidx, nbr = sa_inputs(B, n, m, ns, C, seed=seed + 1)
    rows = sa_rows_ref(xyz, feat, cidx, nbr).double()
    full = mlp_ref(rows, [(w.float().double(), b.float().double()) for w, b in ref])
    exp = full.max(dim=2)[0].reshape(B * m, -1) if pooled else full.reshape(B * m * ns, -1)
    got = _Chain(eng).sa(xyz.cuda(), feat.cuda().contiguous(), cidx.cuda(), nbr.cuda(), C, pooled=pooled)
    err = (got.double().cpu() - exp).abs().max().item()
    if not err <= 2e-5 * (exp.abs().max().item() + 1e-12):
        return f"B={B} n={n} m={m} ns={ns} C={C} dims={dims} pooled={pooled} err={err}"


def f_chain_fp():
    from tests.test_gpu_chain import make_layers, mlp_ref
    from patchaugnet_amd.engine import _Chain
    B, n, m = int(rng.integers(1, 4)), logint(1, 5000), logint(1, 1200)
    c2, c1 = 4 * int(rng.integers(1, 129)), int(rng.choice([0, 3, 4, 5, 64, 256]))
    if B * n * (c2 + c1) > 6e6:
        return None
    dims = _chain_dims(c2 + c1, False)
    seed = int(rng.integers(0, 1 << 30))
    ref, eng = make_layers(dims, seed=seed)
    g = torch.Generator().manual_seed(seed + 1)
    known = torch.randn(B, m, c2, generator=g)
    skip = torch.randn(B, n, c1, generator=g) if c1 else None
    idx3 = torch.randint(0, m, (B, n, 3), generator=g).int()
    w3 = torch.rand(B, n, 3, generator=g)
    w3 = w3 / w3.sum(-1, keepdim=True)
    bi = torch.arange(B)[:, None]
    f = [known[bi, idx3[:, :, t].long()] for t in range(3)]
    interp = (w3[..., 0:1] * f[0] + w3[..., 1:2] * f[1]) + w3[..., 2:3] * f[2]
    rows = torch.cat([interp, skip], dim=-1) if c1 else interp
    exp = mlp_ref(rows.double(), [(w.float().double(), b.float().double()) for w, b in ref]).reshape(B * n, -1)
    got = _Chain(eng).fp(known.cuda(), idx3.cuda(), w3.cuda().contiguous(), skip.cuda() if c1 else None, B, n, m, c2, c1)
    err = (got.double().cpu() - exp).abs().max().item()
    if not err <= 2e-5 * (exp.abs().max().item() + 1e-12):
        return f"B={B} n={n} m={m} c2={c2} c1={c1} dims={dims} err={err}"


def f_sa_mid():
    """the LDS-resident second-level kernel (sa_mid.hip) forced on: 67 -> 64 -> 64 -> n2, 13..20 neighbours, any group count (ragged tiles, several
    tiles per wave), against float64 and against the generic pooled kernel"""
    from tests.test_gpu_chain import make_layers, mlp_ref, sa_inputs, sa_rows_ref
    from patchaugnet_amd.engine import _Chain
    B, n, ns = int(rng.integers(1, 6)), logint(2, 3000), int(rng.integers(13, 21))
    m, n2 = logint(1, min(n, 1500)), 64 * int(rng.integers(1, 5))
    if B * m * ns > 3e5:
        return None
    seed = int(rng.integers(0, 1 << 30))
    ref, eng = make_layers([67, 64, 64, n2], seed=seed)
    xyz, feat, cidx, nbr = sa_inputs(B, n, m, ns, 64, seed=seed + 1)
    rows = sa_rows_ref(xyz, feat, cidx, nbr).double()
    exp = mlp_ref(rows, [(w.float().double(), b.float().double()) for w, b in ref]).max(dim=2)[0].reshape(B * m, -1)
    args = (xyz.cuda(), feat.cuda().contiguous(), cidx.cuda(), nbr.cuda(), 64)
    lib = _lib.lib()
    try:
        lib.pa_chain_mid_enable(1)
        got = _Chain(eng).sa(*args, pooled=True)
        lib.pa_chain_mid_enable(0)
        gen = _Chain(eng).sa(*args, pooled=True)
        torch.cuda.synchronize()
    finally:
        lib.pa_chain_mid_enable(-1)
    scale = exp.abs().max().item() + 1e-12
    err, errg = (got.double().cpu() - exp).abs().max().item(), (got.double() - gen.double()).abs().max().item()
    if not (err <= 2e-5 * scale and errg <= 2e-5 * scale):
        return f"B={B} n={n} m={m} ns={ns} n2={n2} err={err} vs generic={errg}"


def f_sa_tiny():
    """the register-chained first-level kernel (sa_tiny.hip): <= 8 -> 32 -> 32 -> 64, 13..20 neighbours, 1..3 feature channels, any group count (ragged
    last tile, fewer tiles than wavefronts, several tiles per wave), features = coordinates or not; against float64, against the generic pooled kernel
    (equal up to the order of the fp32 additions) and against itself (deterministic)"""
    from tests.test_gpu_chain import make_layers, mlp_ref, sa_inputs, sa_rows_ref
    from patchaugnet_amd.engine import _Chain
    B, n, ns, C = int(rng.integers(1, 40)), logint(2, 5000), int(rng.integers(13, 21)), int(rng.integers(1, 4))
    m = logint(1, min(n, 1200))
    if B * m * ns > 8e5:
        return None
    seed = int(rng.integers(0, 1 << 30))
    ref, eng = make_layers([3 + C, 32, 32, 64], seed=seed)
    xyz, feat, cidx, nbr = sa_inputs(B, n, m, ns, C, seed=seed + 1)
    xyz_d = xyz.cuda()
    feat_d = xyz_d if (C == 3 and rng.integers(0, 2)) else feat.cuda().contiguous()       # the engine's first level passes the coordinates as features
    if feat_d is xyz_d:
        feat = xyz
    rows = sa_rows_ref(xyz, feat, cidx, nbr).double()
    exp = mlp_ref(rows, [(w.float().double(), b.float().double()) for w, b in ref]).max(dim=2)[0].reshape(B * m, -1)
    args = (xyz_d, feat_d, cidx.cuda(), nbr.cuda(), C)
    lib = _lib.lib()
    try:
        lib.pa_chain_tiny_enable(1)
        got = _Chain(eng).sa(*args, pooled=True)
        again = _Chain(eng).sa(*args, pooled=True)
        lib.pa_chain_tiny_enable(0)
        gen = _Chain(eng).sa(*args, pooled=True)
        torch.cuda.synchronize()
    finally:
        lib.pa_chain_tiny_enable(-1)
    scale = exp.abs().max().item() + 1e-12
    err, errg = (got.double().cpu() - exp).abs().max().item(), (got.double() - gen.double()).abs().max().item()
    if not (err <= 2e-5 * scale and errg <= 2e-5 * scale and torch.equal(got, again)):
        return f"B={B} n={n} m={m} ns={ns} C={C} same={feat_d is xyz_d} err={err} vs generic={errg} deterministic={torch.equal(got, again)}"


def f_fpx32():
    """the finest FP level in half-K passes (fpx_f32.hip) forced on / off: it takes launches of >= 65 536 rows (below, the shared-tile tilings run either
    way), 1..4 skip channels, ragged last tiles; the same bits as the 16-row tile kernel, and float64"""
    from tests.test_gpu_chain import make_layers, mlp_ref
    from patchaugnet_amd.engine import _Chain
    B, c1 = int(rng.integers(1, 5)), int(rng.integers(1, 5))
    n = int(rng.integers(65536 // B + 1, 100000 // B + 2)) if rng.integers(0, 4) else logint(1, 30000)
    m = logint(1, 3000)
    seed = int(rng.integers(0, 1 << 30))
    ref, eng = make_layers([256 + c1, 256, 256, 256], seed=seed)
    g = torch.Generator().manual_seed(seed + 1)
    known = torch.randn(B, m, 256, generator=g)
    skip = torch.randn(B, n, c1, generator=g)
    idx3 = torch.randint(0, m, (B, n, 3), generator=g).int()
    w3 = torch.rand(B, n, 3, generator=g)
    w3 = (w3 / w3.sum(-1, keepdim=True)).contiguous()
    args = (known.cuda(), idx3.cuda(), w3.cuda(), skip.cuda(), B, n, m, 256, c1)
    ch = _Chain(eng)
    ch.build_premul(256, c1)
    lib = _lib.lib()
    try:
        lib.pa_chain_fpx32_enable(1)
        got = ch.fp_premul(*args).clone()
        lib.pa_chain_fpx32_enable(0)
        tile = ch.fp_premul(*args).clone()
        torch.cuda.synchronize()
    finally:
        lib.pa_chain_fpx32_enable(-1)
    if not torch.equal(got, tile):
        return f"B={B} n={n} m={m} c1={c1}: half-K kernel and tile kernel differ, max {float((got - tile).abs().max())}"
    sel = torch.randperm(B * n, generator=g)[:4096]                       # float64 reference on a sample of the rows (the whole tensor is up to 100 k x 256)
    bsel, psel = sel // n, sel % n
    interp = sum(w3[bsel, psel, t:t + 1].double() * known.double()[bsel, idx3[bsel, psel, t].long()] for t in range(3))
    exp = mlp_ref(torch.cat([interp, skip[bsel, psel].double()], -1), [(w.float().double(), b.float().double()) for w, b in ref])
    err = (got[sel.cuda()].double().cpu() - exp).abs().max().item()
    if not err <= 2e-5 * (exp.abs().max().item() + 1e-12):
        return f"B={B} n={n} m={m} c1={c1} err={err}"


def f_fpx16():
    """the fp16 path's finest FP level (fpx_f16.hip): both workgroup shapes, fp16 and fp32 pre-multiplied table, any row count; reference with
    the same operand roundings in float64 (2e-3 of the scale: a hidden value next to an fp16 tie may round the other way)"""
    from tests.test_gpu_chain import make_layers
    from patchaugnet_amd.engine import _Chain
    B, n, m, c1 = int(rng.integers(1, 4)), logint(1, 6000), logint(1, 1200), int(rng.integers(1, 5))
    mode, g16 = int(rng.choice([4, 8])), bool(rng.integers(0, 2))
    seed = int(rng.integers(0, 1 << 30))
    ref, eng = make_layers([256 + c1, 256, 256, 256], seed=seed)
    g = torch.Generator().manual_seed(seed + 1)
    known = torch.randn(B, m, 256, generator=g)
    skip = torch.randn(B, n, c1, generator=g)
    idx3 = torch.randint(0, m, (B, n, 3), generator=g).int()
    w3 = torch.rand(B, n, 3, generator=g)
    w3 = (w3 / w3.sum(-1, keepdim=True)).contiguous()
    os.environ["PA_ENGINE_FPX16"] = "1" if g16 else "0"
    ch = _Chain(eng, f16=True)
    ch.build_premul(256, c1)
    os.environ.pop("PA_ENGINE_FPX16")
    lib = _lib.lib()
    lib.pa_fpx16_enable(mode)
    try:
        got = ch.fp_premul(known.cuda(), idx3.cuda(), w3.cuda(), skip.cuda(), B, n, m, 256, c1)
        torch.cuda.synchronize()
    finally:
        lib.pa_fpx16_enable(-1)
    h16 = lambda t: t.half().double()
    (w1, b1), (w2, b2), (w3_, b3) = [(w.float().double(), b.float().double()) for w, b in ref]
    gk = (h16(known.double()) @ h16(w1[:, :256]).t()).float().double()
    if g16:
        gk = h16(gk)
    bi = torch.arange(B)[:, None]
    interp = sum(w3[..., t:t + 1].double() * gk[bi, idx3[:, :, t].long()] for t in range(3))
    h1 = torch.relu(interp + skip.double() @ w1[:, 256:].t() + b1)
    h2 = torch.relu(h16(h1) @ h16(w2).t() + b2)
    exp = torch.relu(h16(h2) @ h16(w3_).t() + b3).reshape(B * n, -1)
    err = (got.double().cpu() - exp).abs().max().item()
    if not err <= 2e-3 * (exp.abs().max().item() + 1e-12):
        return f"B={B} n={n} m={m} c1={c1} mode={mode} g16={g16} err={err}"


def f_fpx3():
    """opt-in "f32x3" (fpx_f32x3.hip: the finest FP level and its pre-multiply from (hi, lo) fp16 operand pairs) against float64 at the FP32 families'
    tolerance, 2e-5 of the scale; any row count, weights of random magnitude (the per-layer power-of-two scale)"""
    from tests.test_gpu_chain import make_layers, mlp_ref
    from patchaugnet_amd.engine import _Chain
    B, n, m, c1 = int(rng.integers(1, 4)), logint(1, 6000), logint(1, 1200), int(rng.integers(1, 5))
    seed = int(rng.integers(0, 1 << 30))
    ref, eng = make_layers([256 + c1, 256, 256, 256], seed=seed)
    mag = float(2.0 ** rng.integers(-6, 7))                       # layer weights from 2^-6 to 2^6 times the usual
    ref = [(w * (mag if i == 1 else 1.0), b) for i, (w, b) in enumerate(ref)]
    eng = [((wt * (mag if i == 1 else 1.0)).contiguous(), b, k, kp, nn) for i, (wt, b, k, kp, nn) in enumerate(eng)]
    g = torch.Generator().manual_seed(seed + 1)
    known = torch.randn(B, m, 256, generator=g) * float(2.0 ** rng.integers(-3, 6))
    skip = torch.randn(B, n, c1, generator=g)
    idx3 = torch.randint(0, m, (B, n, 3), generator=g).int()
    w3 = torch.rand(B, n, 3, generator=g)
    w3 = (w3 / w3.sum(-1, keepdim=True)).contiguous()
    ch = _Chain(eng)
    ch.build_premul(256, c1, x3=True)
    got = ch.fp_premul(known.cuda(), idx3.cuda(), w3.cuda(), skip.cuda(), B, n, m, 256, c1)
    bi = torch.arange(B)[:, None]
    interp = sum(w3[..., t:t + 1].double() * known.double()[bi, idx3[:, :, t].long()] for t in range(3))
    exp = mlp_ref(torch.cat([interp, skip.double()], -1), [(w.float().double(), b.float().double()) for w, b in ref]).reshape(B * n, -1)
    err = (got.double().cpu() - exp).abs().max().item()
    if not err <= 2e-5 * (exp.abs().max().item() + 1e-12):
        return f"B={B} n={n} m={m} c1={c1} mag={mag} err={err} scale={exp.abs().max().item()}"


def f_attention_f16():
    """the fp16 attention (attention_f16.hip: hi / lo logits, column-online scaling, fused layer behind it where the engine fuses) against the
    fp32 oracle statement: cosine of every point's output row >= 0.999 and max error <= 2e-2 of the scale"""
    from patchaugnet_amd import backbone
    from patchaugnet_amd.engine import _Attn
    from oracle import models_cpu
    b, n, c = int(rng.integers(1, 4)), logint(1, 1100), int(rng.choice([64, 128, 256]))
    if n * c > 300000:
        n = max(1, 300000 // c)
    torch.manual_seed(int(rng.integers(0, 1 << 30)))
    sa = backbone.SALayer(c, 8).eval()
    for p in sa.parameters():
        p.data.mul_(0.5)
    x = torch.randn(b, c, n)
    sd = {"s." + k: v for k, v in sa.state_dict().items()}
    with torch.no_grad():
        ref = models_cpu.sa_layer(sd, "s", x, 8)
        xm = x.transpose(1, 2).contiguous().view(b * n, c).cuda()
        got = _Attn(sa, xm.device, f16=True).run(xm, b, n).view(b, n, c).transpose(1, 2).cpu()
    err = (got - ref).abs().max().item()
    cos = torch.nn.functional.cosine_similarity(got.double(), ref.double(), dim=1).min().item()
    if not (err <= 2e-2 * max(ref.abs().max().item(), 1.0) and cos >= 0.999):
        return f"b={b} n={n} c={c} err={err} cos={cos}"


def _seed_module(m, seed):
    from patchaugnet_amd.weights import seeded_state_dict
    m.load_state_dict(seeded_state_dict(m.state_dict(), seed=seed))
    return m.cuda().eval()


def f_netvlad():
    from patchaugnet_amd import loupe
    from patchaugnet_amd.engine import _Vlad
    b, n, k = int(rng.integers(1, 5)), logint(1, 5000), int(rng.integers(1, 65))
    v = _seed_module(loupe.NetVLADBase(256, n, k, 256, gating=False), seed=int(rng.integers(0, 1 << 30)))
    x = torch.randn(b, n, 256, device="cuda") * 0.7
    pad_l, pad_r = int(rng.integers(0, 4)), int(rng.integers(0, 4))
    with torch.no_grad():
        ref = v(x.transpose(1, 2).unsqueeze(-1))
        out = torch.full((b, 256, pad_l + k + pad_r), 7.0, device="cuda")
        _Vlad(v, x.device).run(x, out, pad_l + k + pad_r, pad_l)
        out_t = torch.full((b, pad_l + k + pad_r, 256), 7.0, device="cuda")
        _Vlad(v, x.device).run(x, out_t, pad_l + k + pad_r, pad_l, rows=True)
    got = out[:, :, pad_l:pad_l + k]
    err = (got - ref).abs().max().item()
    clean = torch.all(out[:, :, :pad_l] == 7.0) and torch.all(out[:, :, pad_l + k:] == 7.0) and torch.all(out_t[:, :pad_l] == 7.0) and torch.all(out_t[:, pad_l + k:] == 7.0)
    if not (err <= 3e-5 and clean and torch.equal(out_t[:, pad_l:pad_l + k].transpose(1, 2), got)):
        return f"b={b} n={n} k={k} pads=({pad_l},{pad_r}) err={err} clean={bool(clean)}"


def f_afa():
    from patchaugnet_amd import loupe
    from patchaugnet_amd.engine import _Afa
    b, ktot = logint(1, 150), int(rng.integers(1, 130))
    afa = _seed_module(loupe.AdaptiveFeatureAggregator(256, ktot, 256), seed=int(rng.integers(0, 1 << 30)))
    v = torch.nn.functional.normalize(torch.randn(b, 256, ktot, device="cuda"), dim=1)
    with torch.no_grad():
        ref = afa(v).squeeze(-1)
        eng = _Afa(afa, v.device)
        got = eng.run(v.contiguous())
        got_rows = eng.run_rows(v.transpose(1, 2).contiguous())
        got_fused = eng.run_fused(v.transpose(1, 2).contiguous())          # the two-launch head of the shipped engine
    e1, e2, e3 = (got - ref).abs().max().item(), (got_rows - ref).abs().max().item(), (got_fused - ref).abs().max().item()
    if not (e1 <= 3e-5 and e2 <= 3e-5 and e3 <= 3e-5):
        return f"b={b} ktot={ktot} err={e1} err_rows={e2} err_fused={e3}"


def f_linear_lds():
    """pa_linear's k = 256 kernel with the weight half resident in LDS (linear_lds.hip): bit-identical to the chain kernel on any row count, row
    strides, widths that are multiples of 128, bias / ReLU / residual; fp64 check on top"""
    import ctypes
    lib = _lib.lib()
    lib.pa_linear_lds_enable.argtypes, lib.pa_linear_lds_enable.restype = [ctypes.c_int], None
    rows, n = logint(1, 40000), 128 * int(rng.integers(1, 6))
    ldx = 256 + 4 * int(rng.integers(0, 3)) * int(rng.integers(0, 12))
    relu, res = int(rng.integers(0, 2)), bool(rng.integers(0, 2))
    ldr = n + 4 * int(rng.integers(0, 5)) if res else 0
    x = torch.randn(rows, ldx, device="cuda")
    wt = torch.randn(256, n, device="cuda") / 16
    bias = torch.randn(n, device="cuda")
    r = torch.randn(rows, ldr, device="cuda") if res else None
    outs = []
    try:
        for on in (0, 1):
            lib.pa_linear_lds_enable(on)
            out = torch.full((rows, n), float("nan"), device="cuda")
            call("pa_linear", rows, 256, n, ptr(x), ldx, ptr(wt), ptr(pack_weights(wt)), ptr(bias), relu, ptr(r), ldr, ptr(out), n)
            outs.append(out)
    finally:
        lib.pa_linear_lds_enable(-1)
    ref = x[:, :256].double() @ wt.double() + bias.double()
    if relu:
        ref = ref.clamp_min(0)
    if res:
        ref = ref + r[:, :n].double()
    err = (outs[1].double() - ref).abs().max().item()
    if not (torch.equal(outs[0], outs[1]) and err <= 2e-5 * max(ref.abs().max().item(), 1.0)):
        return f"rows={rows} n={n} ldx={ldx} relu={relu} res={res} ldr={ldr} identical={torch.equal(outs[0], outs[1])} err={err}"


def _grad_check(fn_hip, fn_ref, inputs, tol=3e-5):
    """value and every input gradient of an autograd function on the HIP kernels against the same torch statement in fp64 on the device"""
    hin = [t.clone().requires_grad_(True) for t in inputs]
    rin = [t.double().requires_grad_(True) for t in inputs]
    oh, orf = fn_hip(*hin), fn_ref(*rin)
    w = torch.randn_like(orf)
    (oh * w.float()).sum().backward()
    (orf * w).sum().backward()
    # error relative to the tensor's scale, floored at 1 like the other GEMM families (a gradient that is zero in exact arithmetic has no scale)
    worst = ((oh.double() - orf).abs().max() / orf.abs().max().clamp_min(1.0)).item()
    for a, b in zip(hin, rin):
        worst = max(worst, ((a.grad.double() - b.grad).abs().max() / b.grad.abs().max().clamp_min(1.0)).item())
    return None if worst <= tol else worst


def f_train_glue():
    """csrc/train_glue.hip under autograd: NetVLAD tail, dim-1 normalise, APFA attention, BatchNorm1d rows -- random shapes, fp64 torch on the device"""
    import torch.nn.functional as F
    from patchaugnet_amd import train_ops
    which = int(rng.integers(0, 4))
    if which == 0:
        b, c, k, n = int(rng.integers(1, 5)), 4 * int(rng.integers(1, 65)), int(rng.choice([1, 2, 4, 7, 16, 33, 64, 100])), logint(1, 5000)
        pre, x, cw2 = torch.randn(b, k, n, device="cuda") * 2, torch.randn(b, c, n, device="cuda"), torch.randn(1, c, k, device="cuda") / c ** 0.5

        def ref(pre, x, cw2):
            act = torch.softmax(pre, dim=1)
            return F.normalize(torch.matmul(x, act.transpose(1, 2)) - act.sum(-1).unsqueeze(1) * cw2, dim=1)
        bad = _grad_check(train_ops.netvlad_tail, ref, [pre, x, cw2])
        return None if bad is None else f"netvlad_tail b={b} c={c} k={k} n={n} rel={bad}"
    if which == 1:
        b, c, m = int(rng.integers(1, 20)), logint(1, 600), int(rng.choice([1, 1, 3, 31, 32, 33, 1000]))
        x = torch.randn(b, c, m, device="cuda") if m > 1 else torch.randn(b, c, device="cuda")
        bad = _grad_check(train_ops.l2_normalize, lambda t: F.normalize(t, dim=1), [x])
        return None if bad is None else f"l2_normalize b={b} c={c} m={m} rel={bad}"
    if which == 2:
        b, c, k = int(rng.integers(1, 20)), logint(1, 300), logint(1, 1200)
        x, r = torch.randn(b, c, k, device="cuda"), torch.randn(b, c, k, device="cuda")

        def ref(x, r):
            return F.relu(x + x * torch.softmax(r.max(dim=1)[0], dim=-1).unsqueeze(1))
        bad = _grad_check(train_ops.afa_attention, ref, [x, r])
        return None if bad is None else f"afa_attention b={b} c={c} k={k} rel={bad}"
    r_, f_ = int(rng.integers(4, 64)), logint(1, 1500)
    bn_h = torch.nn.BatchNorm1d(f_).cuda()
    with torch.no_grad():
        bn_h.weight.uniform_(0.5, 1.5)
        bn_h.bias.normal_(0, 0.2)
    bn_r = torch.nn.BatchNorm1d(f_).cuda().double()
    bn_r.load_state_dict({k: (v.double() if v.is_floating_point() else v) for k, v in bn_h.state_dict().items()})
    x = torch.randn(r_, f_, device="cuda") * 2 + 0.5
    bad = _grad_check(lambda t: train_ops.bn_rows(bn_h, t, True), lambda t: bn_r(t), [x], tol=2e-4)
    ok = bad is None and torch.allclose(bn_h.running_var.double(), bn_r.running_var, rtol=1e-5, atol=1e-6) and int(bn_h.num_batches_tracked) == 1
    return None if ok else f"bn_rows r={r_} f={f_} rel={bad}"


FAMILIES = (("fps", f_fps), ("knn", f_knn), ("3nn", f_3nn), ("knn_grid", f_knn_grid), ("3nn_grid", f_3nn_grid), ("gather", f_gather), ("backward", f_backward), ("linear", f_linear),
            ("attention", f_attention), ("chain_sa", f_chain_sa), ("chain_fp", f_chain_fp), ("netvlad", f_netvlad), ("afa", f_afa), ("linear_lds", f_linear_lds), ("train_glue", f_train_glue),
            ("sa_mid", f_sa_mid), ("sa_tiny", f_sa_tiny), ("fpx32", f_fpx32), ("fpx16", f_fpx16), ("attention_f16", f_attention_f16), ("fpx3", f_fpx3))

if __name__ == "__main__":
    if len(sys.argv) > 1:
        BUDGET = float(sys.argv[1])
    ok = True
    fams = FAMILIES
    only = os.environ.get("FUZZ_ONLY")
    for name, fn in fams:
        if only and name not in only.split(","):
            continue
        try:
            ok = run(name, fn) and ok
        except Exception as ex:
            print(f"EXC  {name}: {ex!r}", flush=True)
            ok = False
    sys.exit(0 if ok else 1)
