"""pa_mlp_chain (fused gather / interpolate + shared-MLP + max-pool on MFMA) against a float64 torch restatement of the
unfused reference sequence (pointops.py:559-570, pt_util.py:16-41, patch_aug_net.py:236, :354-359).
Tolerance: fp32 MFMA is an exact k-ordered fmaf chain, so only summation order differs from fp64: rtol 2e-5."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def make_layers(dims, seed):
    """Random folded layers: list of (W (n,k), b (n)) in float64 + engine-format tensors."""
    g = torch.Generator().manual_seed(seed)
    ref, eng = [], []
    for k, n in zip(dims[:-1], dims[1:]):
        w = torch.randn(n, k, generator=g, dtype=torch.float64) * (2.0 / k) ** 0.5
        b = torch.randn(n, generator=g, dtype=torch.float64) * 0.1
        kpad = (k + 3) // 4 * 4
        wt = torch.zeros(kpad, n, dtype=torch.float64)
        wt[:k] = w.t()
        ref.append((w, b))
        eng.append((wt.float().cuda().contiguous(), b.float().cuda().contiguous(), k, kpad, n))
    return ref, eng


def mlp_ref(x, layers):
    for w, b in layers:
        x = torch.relu(x @ w.t() + b)
    return x


def close(got, ref, rtol=2e-5):
    got, ref = got.double().cpu(), ref.double().cpu()
    scale = ref.abs().max().item() + 1e-12
    err = (got - ref).abs().max().item()
    assert err <= rtol * scale, f"max err {err:.3e} vs scale {scale:.3e}"


@pytest.mark.parametrize("rows,dims", [(1000, [64, 64, 256]), (4096, [259, 256, 256, 256]), (33, [7, 16]), (5000, [768, 256, 256]),
                                       (640, [320, 256, 256]), (96, [256, 512]), (200, [12, 32, 32, 64])])
def test_plain_rows(rows, dims):
    from patchaugnet_amd.engine import _Chain
    ref, eng = make_layers(dims, seed=rows)
    x = torch.randn(rows, dims[0], dtype=torch.float64)
    got = _Chain(eng).plain(x.float().cuda().contiguous())
    close(got, mlp_ref(x.float().double(), [(w.float().double(), b.float().double()) for w, b in ref]))


def sa_inputs(B, n, m, ns, C, seed):
    g = torch.Generator().manual_seed(seed)
    xyz = torch.rand(B, n, 3, generator=g) * 2 - 1
    feat = xyz.clone() if C == 3 else torch.randn(B, n, C, generator=g)
    cidx = torch.stack([torch.randperm(n, generator=g)[:m] for _ in range(B)]).int()
    nbr = torch.randint(0, n, (B, m, ns), generator=g).int()
    return xyz, feat, cidx, nbr


def sa_rows_ref(xyz, feat, cidx, nbr):
    B, m, ns = nbr.shape
    bi = torch.arange(B)[:, None, None]
    p_xyz, p_f = xyz[bi, nbr.long()], feat[bi, nbr.long()]                   # (B, m, ns, .)
    c_xyz, c_f = xyz[bi[:, :, 0], cidx.long()], feat[bi[:, :, 0], cidx.long()]
    return torch.cat([p_xyz - c_xyz[:, :, None], p_f - c_f[:, :, None]], dim=-1)  # (B, m, ns, 3+C) fp32 subtraction


@pytest.mark.parametrize("B,n,m,ns,C,dims", [(2, 4096, 1024, 20, 3, [6, 32, 32, 64]), (3, 1024, 128, 20, 64, [67, 64, 64, 256]),
                                              (2, 100, 13, 20, 64, [67, 64, 64, 256]), (1, 64, 9, 16, 8, [11, 16, 32]),
                                              (2, 50, 6, 32, 3, [6, 32, 48]), (2, 300, 30, 17, 5, [8, 64, 16])])
def test_sa_pooled(B, n, m, ns, C, dims):
    from patchaugnet_amd.engine import _Chain
    ref, eng = make_layers(dims, seed=n + ns)
    xyz, feat, cidx, nbr = sa_inputs(B, n, m, ns, C, seed=m)
    rows = sa_rows_ref(xyz, feat, cidx, nbr).double()
    exp = mlp_ref(rows, [(w.float().double(), b.float().double()) for w, b in ref]).max(dim=2)[0].reshape(B * m, -1)
    got = _Chain(eng).sa(xyz.cuda(), feat.cuda().contiguous(), cidx.cuda(), nbr.cuda(), C, pooled=True)
    close(got, exp)


@pytest.mark.parametrize("B,n,m,ns,C,dims", [(2, 128, 16, 20, 256, [259, 256, 256, 512]), (1, 40, 5, 7, 4, [7, 32, 16])])
def test_sa_unpooled_then_rowgroup_max(B, n, m, ns, C, dims):
    from patchaugnet_amd import _lib
    from patchaugnet_amd.engine import _Chain
    ref, eng = make_layers(dims, seed=7)
    xyz, feat, cidx, nbr = sa_inputs(B, n, m, ns, C, seed=3)
    rows = sa_rows_ref(xyz, feat, cidx, nbr).double()
    full = mlp_ref(rows, [(w.float().double(), b.float().double()) for w, b in ref])
    got = _Chain(eng).sa(xyz.cuda(), feat.cuda().contiguous(), cidx.cuda(), nbr.cuda(), C, pooled=False)
    close(got, full.reshape(B * m * ns, -1))
    pooled = torch.empty(B * m, dims[-1], device="cuda")
    _lib.call("pa_rowgroup_max", B * m, ns, dims[-1], _lib.ptr(got), _lib.ptr(pooled))
    assert torch.equal(pooled.cpu(), got.view(B * m, ns, -1).max(dim=1)[0].cpu())


@pytest.mark.parametrize("f16", [False, True])
@pytest.mark.parametrize("B,n,m,ns,C,dims", [(32, 256, 64, 20, 256, [259, 256, 256, 512]), (70, 128, 16, 20, 256, [259, 256, 256, 512]),
                                              (20, 256, 64, 20, 128, [131, 128, 128, 256]), (65, 64, 16, 18, 61, [64, 64, 256, 128])])
def test_sa_pooled_wide_hidden_equals_unpooled(B, n, m, ns, C, dims, f16):
    """Wide hidden layers: the shared-tile pooled kernel (the engine picks it for 256 <= four-group tiles < 2048; any count below 2048
    on the fp16 path) must give the same bits as the unpooled kernel followed by pa_rowgroup_max (what other batch sizes run) --
    results may not depend on the batch size."""
    from patchaugnet_amd import _lib
    from patchaugnet_amd.engine import _Chain
    ref, eng = make_layers(dims, seed=11)
    xyz, feat, cidx, nbr = sa_inputs(B, n, m, ns, C, seed=5)
    chain = _Chain(eng, f16=f16)
    assert chain.pooled_ok(B * m, ns) and not chain.hidden_ok_pooled
    assert not chain.pooled_ok(8192, ns) and chain.pooled_ok(8, ns) == f16
    args = (xyz.cuda(), feat.cuda().contiguous(), cidx.cuda(), nbr.cuda(), C)
    got = chain.sa(*args, pooled=True)
    full = chain.sa(*args, pooled=False)
    two_step = torch.empty(B * m, dims[-1], device="cuda")
    _lib.call("pa_rowgroup_max", B * m, ns, dims[-1], _lib.ptr(full), _lib.ptr(two_step))
    assert torch.equal(got, two_step)
    if f16:
        return
    rows = sa_rows_ref(xyz, feat, cidx, nbr).double()
    exp = mlp_ref(rows, [(w.float().double(), b.float().double()) for w, b in ref]).max(dim=2)[0].reshape(B * m, -1)
    close(got, exp)


@pytest.mark.parametrize("B,n,m,c2,c1,dims", [(2, 128, 16, 512, 256, [768, 256, 256]), (2, 1024, 128, 256, 64, [320, 256, 256]),
                                               (2, 4096, 1024, 256, 3, [259, 256, 256, 256]), (1, 50, 7, 8, 0, [8, 16]), (1, 33, 9, 4, 5, [9, 32, 16])])
def test_fp_interpolate(B, n, m, c2, c1, dims):
    from patchaugnet_amd.engine import _Chain
    ref, eng = make_layers(dims, seed=c2)
    g = torch.Generator().manual_seed(n)
    known = torch.randn(B, m, c2, generator=g)
    skip = torch.randn(B, n, c1, generator=g) if c1 else None
    idx3 = torch.randint(0, m, (B, n, 3), generator=g).int()
    w3 = torch.rand(B, n, 3, generator=g)
    w3 = w3 / w3.sum(-1, keepdim=True)
    bi = torch.arange(B)[:, None]
    f = [known[bi, idx3[:, :, t].long()] for t in range(3)]
    interp = (w3[..., 0:1] * f[0] + w3[..., 1:2] * f[1]) + w3[..., 2:3] * f[2]            # fp32, reference order
    rows = torch.cat([interp, skip], dim=-1) if c1 else interp
    exp = mlp_ref(rows.double(), [(w.float().double(), b.float().double()) for w, b in ref]).reshape(B * n, -1)
    got = _Chain(eng).fp(known.cuda(), idx3.cuda(), w3.cuda().contiguous(), skip.cuda() if c1 else None, B, n, m, c2, c1)
    close(got, exp)


@pytest.mark.parametrize("B,n,m,ns,C,dims", [(32, 128, 16, 20, 256, [259, 256, 256, 512]), (2, 128, 16, 20, 256, [259, 256, 256, 512]),
                                              (5, 256, 37, 16, 128, [131, 128, 128, 256]), (3, 64, 7, 32, 61, [64, 64, 256, 128]),
                                              (1, 40, 3, 19, 61, [64, 64, 64])])
def test_sa_atomic_pooled_epilogue_equals_unpooled_then_max(B, n, m, ns, C, dims):
    """pooled = 2: the unpooled 16-row tiling with the max over the neighbourhood folded into the last layer's epilogue (masked DPP-row maxima +
    integer atomicMax on the non-negative float patterns) gives the SAME BITS as writing every row and running pa_rowgroup_max -- group
    boundaries inside a row tile (ns = 20, 19), exactly on it (16, 32), and a ragged last tile."""
    from patchaugnet_amd import _lib
    from patchaugnet_amd.engine import _Chain
    ref, eng = make_layers(dims, seed=13)
    xyz, feat, cidx, nbr = sa_inputs(B, n, m, ns, C, seed=9)
    chain = _Chain(eng)
    assert chain.atomic_pool_ok(B * m, ns)
    args = (xyz.cuda(), feat.cuda().contiguous(), cidx.cuda(), nbr.cuda(), C)
    full = chain.sa(*args, pooled=False)
    two_step = torch.empty(B * m, dims[-1], device="cuda")
    _lib.call("pa_rowgroup_max", B * m, ns, dims[-1], _lib.ptr(full), _lib.ptr(two_step))
    for _ in range(2):                      # the second call finds a dirty output buffer: the launcher zero-fills it
        got = chain.sa(*args, pooled=2)
        assert got.shape == two_step.shape and torch.equal(got, two_step)


@pytest.mark.parametrize("B,n,m,ns,C", [(8, 4096, 1024, 20, 3), (3, 1024, 250, 20, 3), (2, 333, 41, 17, 3), (5, 512, 128, 16, 3), (1, 100, 9, 13, 2),
                                         (32, 4096, 1024, 20, 3)])
def test_sa_first_level_persistent_register_chained_kernel(B, n, m, ns, C):
    """sa_tiny.hip (round 6: persistent workgroups, the level's weights in registers, every layer fed straight from the previous layer's MFMA
    accumulators, gather pipelined across tiles) on the first level's shape <= 8 -> 32 -> 32 -> 64: against float64 and against the generic pooled
    chain kernel -- equal up to the ORDER of the fp32 additions inside a dot product (k-steps follow the accumulator layout), not bit for bit;
    deterministic from launch to launch.  Ragged group counts (the last tile partly empty, fewer tiles than wavefronts), nsample 13 .. 20 (both
    row-tile counts), two feature channels."""
    import ctypes
    from patchaugnet_amd import _lib
    from patchaugnet_amd.engine import _Chain
    lib = _lib.lib()
    lib.pa_chain_tiny_enable.argtypes, lib.pa_chain_tiny_enable.restype = [ctypes.c_int], None
    ref, eng = make_layers([3 + C, 32, 32, 64], seed=17)
    xyz, feat, cidx, nbr = sa_inputs(B, n, m, ns, C, seed=21)
    chain = _Chain(eng)
    args = (xyz.cuda(), feat.cuda().contiguous(), cidx.cuda(), nbr.cuda(), C)
    try:
        lib.pa_chain_tiny_enable(0)
        generic = chain.sa(*args, pooled=True)
        lib.pa_chain_tiny_enable(1)
        tiny = chain.sa(*args, pooled=True)
        tiny2 = chain.sa(*args, pooled=True)
        lib.pa_chain_tiny_enable(-1)
        default = chain.sa(*args, pooled=True)
    finally:
        lib.pa_chain_tiny_enable(-1)
    assert torch.equal(tiny, tiny2)
    assert torch.equal(default, tiny), "the default rule must pick this kernel for every batch size of the shape (results must not depend on batching)"
    rows = sa_rows_ref(xyz, feat, cidx, nbr).double()
    exp = mlp_ref(rows, [(w.float().double(), b.float().double()) for w, b in ref]).max(dim=2)[0].reshape(B * m, -1)
    close(tiny, exp)
    close(generic, exp)
    close(tiny, generic.double(), rtol=2e-5)


@pytest.mark.parametrize("B,n,m,c1", [(3, 4096, 1024, 3), (1, 1000, 256, 3), (2, 77, 19, 1), (1, 300, 64, 4)])
def test_finest_fp_level_from_split_fp16_operands_meets_the_fp32_tolerance(B, n, m, c1):
    """Opt-in "f32x3" (fpx_f32x3.hip): every product of the two 256 -> 256 layers as hi(a) hi(w) + lo(a) hi(w) + hi(a) lo(w) on the fp16 MFMA
    (weights scaled by a power of two per layer).  It has to pass the SAME test as the fp32 kernels -- 2e-5 of the tensor's scale against
    float64 -- and is reported next to the exact-fp32 kernel's error on the same data."""
    from patchaugnet_amd.engine import _Chain
    c2 = 256
    ref, eng = make_layers([c2 + c1, 256, 256, 256], seed=21 + c1)
    g = torch.Generator().manual_seed(n)
    known = torch.randn(B, m, c2, generator=g)
    skip = torch.randn(B, n, c1, generator=g)
    idx3 = torch.randint(0, m, (B, n, 3), generator=g).int()
    w3 = torch.rand(B, n, 3, generator=g)
    w3 = (w3 / w3.sum(-1, keepdim=True)).contiguous()
    bi = torch.arange(B)[:, None]
    interp = sum(w3[..., t:t + 1].double() * known.double()[bi, idx3[:, :, t].long()] for t in range(3))
    exp = mlp_ref(torch.cat([interp, skip.double()], -1), [(w.float().double(), b.float().double()) for w, b in ref]).reshape(B * n, -1)
    args = (known.cuda(), idx3.cuda(), w3.cuda(), skip.cuda(), B, n, m, c2, c1)
    ch3 = _Chain(eng)
    ch3.build_premul(c2, c1, x3=True)
    assert ch3._premul["x3"] is not None
    got = ch3.fp_premul(*args)
    ch = _Chain(eng)
    ch.build_premul(c2, c1)
    exact = ch.fp_premul(*args)
    torch.cuda.synchronize()
    scale = exp.abs().max().item()
    e3, e1 = (got.double().cpu() - exp).abs().max().item() / scale, (exact.double().cpu() - exp).abs().max().item() / scale
    print(f"relative max error against float64: split fp16 operands {e3:.2e}, fp32 MFMA {e1:.2e}")
    close(got, exp)
    close(exact, exp)
    assert e3 <= 4e-6


@pytest.mark.parametrize("B,n,m,c1", [(32, 4096, 1024, 3), (3, 4096, 1024, 3), (1, 1000, 256, 3), (2, 77, 19, 1), (5, 300, 64, 4), (16, 4096, 1024, 3)])
def test_finest_fp_level_in_half_k_passes_is_bit_identical_to_the_tile_kernel(B, n, m, c1):
    """fpx_f32.hip (round 6: the 16 x 256 activation tile as two 16 x 128 halves through 9 KB of LDS per wavefront, the second layer's input kept in
    registers, twelve wavefronts per workgroup) against the wave-private 16-row tile kernel of pa_chain_kernel.h on the finest level's shape
    (xyz skip, 256 -> 256 -> 256): same fp32 MFMA with k ascending -> the same bits, ragged last tiles and 1 / 3 / 4 skip channels included; and
    against float64."""
    import ctypes
    from patchaugnet_amd import _lib
    from patchaugnet_amd.engine import _Chain
    lib = _lib.lib()
    lib.pa_chain_fpx32_enable.argtypes, lib.pa_chain_fpx32_enable.restype = [ctypes.c_int], None
    c2 = 256
    ref, eng = make_layers([c2 + c1, 256, 256, 256], seed=31 + c1)
    g = torch.Generator().manual_seed(n + B)
    known = torch.randn(B, m, c2, generator=g)
    skip = torch.randn(B, n, c1, generator=g)
    idx3 = torch.randint(0, m, (B, n, 3), generator=g).int()
    w3 = torch.rand(B, n, 3, generator=g)
    w3 = (w3 / w3.sum(-1, keepdim=True)).contiguous()
    args = (known.cuda(), idx3.cuda(), w3.cuda(), skip.cuda(), B, n, m, c2, c1)
    ch = _Chain(eng)
    ch.build_premul(c2, c1)
    outs = []
    try:
        for on in (0, 1, 1):
            lib.pa_chain_fpx32_enable(on)
            outs.append(ch.fp_premul(*args).clone())
    finally:
        lib.pa_chain_fpx32_enable(-1)
    torch.cuda.synchronize()
    if B * n >= 30000:          # the tile kernel's 16-row wave-private form (what the half-K kernel replaces) runs from 30 000 rows; below, the shared-tile tilings
        assert torch.equal(outs[1], outs[0])
    assert torch.equal(outs[2], outs[1])
    bi = torch.arange(B)[:, None]
    interp = sum(w3[..., t:t + 1].double() * known.double()[bi, idx3[:, :, t].long()] for t in range(3))
    exp = mlp_ref(torch.cat([interp, skip.double()], -1), [(w.float().double(), b.float().double()) for w, b in ref]).reshape(B * n, -1)
    close(outs[1], exp)
    close(outs[0], exp)


@pytest.mark.parametrize("B,n,m,ns,n2", [(32, 1024, 128, 20, 256), (40, 1024, 256, 20, 128), (3, 300, 50, 17, 256), (2, 100, 13, 16, 64), (1, 64, 3, 13, 128)])
def test_sa_second_level_lds_resident_kernel(B, n, m, ns, n2):
    """sa_mid.hip (weights of 67 -> 64 -> 64 -> n2 resident in LDS, activations in registers, a wave per 4-group tile): against float64, and
    against the generic pooled kernel -- equal up to the order of the fp32 additions inside a dot product, not bit for bit.  Cases: the model's
    shapes (one tile per wave at B = 32; PPT-Net's 256 centres -> the persistent loop runs several tiles per wave), 17 / 16 / 13 neighbours
    (padding rows, the four-row-tile build), a ragged last tile."""
    from patchaugnet_amd import _lib
    from patchaugnet_amd.engine import _Chain
    dims = [67, 64, 64, n2]
    ref, eng = make_layers(dims, seed=n + ns)
    xyz, feat, cidx, nbr = sa_inputs(B, n, m, ns, 64, seed=m)
    rows = sa_rows_ref(xyz, feat, cidx, nbr).double()
    exp = mlp_ref(rows, [(w.float().double(), b.float().double()) for w, b in ref]).max(dim=2)[0].reshape(B * m, -1)
    args = (xyz.cuda(), feat.cuda().contiguous(), cidx.cuda(), nbr.cuda(), 64)
    lib = _lib.lib()
    try:
        lib.pa_chain_mid_enable(1)
        got = _Chain(eng).sa(*args, pooled=True)
        lib.pa_chain_mid_enable(0)
        generic = _Chain(eng).sa(*args, pooled=True)
        torch.cuda.synchronize()
    finally:
        lib.pa_chain_mid_enable(-1)
    close(got, exp)
    close(generic, exp)
    close(got, generic.double(), rtol=2e-5)
    assert not torch.equal(got, generic) or B * m < 64          # a different kernel did run (different summation order)


@pytest.mark.parametrize("rows,n,relu,res,ldx", [(4096, 256, 0, False, 256), (32768, 256, 0, False, 256), (5003, 512, 1, True, 300), (17, 128, 1, False, 256),
                                                  (2048, 256, 1, True, 256)])
def test_linear_with_lds_resident_weights_is_bit_identical_to_the_chain_kernel(rows, n, relu, res, ldx):
    """pa_linear's k = 256 kernel with a 128-column weight half resident in LDS (linear_lds.hip) performs the chain kernel's arithmetic:
    same fp32 MFMA, k ascending, bias, activation, residual -> identical bits, ragged last tile and strided rows included."""
    import ctypes
    from patchaugnet_amd import _lib
    from patchaugnet_amd._lib import call, ptr
    lib = _lib.lib()
    lib.pa_linear_lds_enable.argtypes, lib.pa_linear_lds_enable.restype = [ctypes.c_int], None
    g = torch.Generator().manual_seed(rows + n)
    k = 256
    x = torch.randn(rows, ldx, generator=g).cuda()
    wt = (torch.randn(k, n, generator=g) * (2.0 / k) ** 0.5).cuda()
    bias = (torch.randn(n, generator=g) * 0.1).cuda()
    resid = torch.randn(rows, n, generator=g).cuda() if res else None
    outs = []
    try:
        for on in (0, 1, 1):
            lib.pa_linear_lds_enable(on)
            out = torch.full((rows, n), float("nan"), device="cuda")
            call("pa_linear", rows, k, n, ptr(x), ldx, ptr(wt), None, ptr(bias), relu, ptr(resid) if res else None, n if res else 0, ptr(out), n)
            outs.append(out)
    finally:
        lib.pa_linear_lds_enable(-1)
    torch.cuda.synchronize()
    assert torch.equal(outs[1], outs[0]) and torch.equal(outs[2], outs[0])
    exp = x[:, :k].double() @ wt.double() + bias.double()
    if relu:
        exp = torch.relu(exp)
    if res:
        exp = exp + resid.double()
    close(outs[1], exp)
