"""Parity of the HIP kernels (through the C ABI / patchaugnet_amd.pointops) with the CPU oracle.
Bit-exact for every index output and every pure gather; K10 bit-exact given the fixed summation order."""
import numpy as np
import pytest
import torch

from oracle import oracle_ops as o

pytestmark = pytest.mark.gpu
RNG = np.random.default_rng(11)


@pytest.fixture(scope="module")
def P():
    from patchaugnet_amd import pointops
    return pointops


def cloud(b, n, kind="uniform"):
    x = (RNG.random((b, n, 3), dtype=np.float32) * 2 - 1).astype(np.float32)
    if kind == "lattice":      # coarse lattice => many exact distance ties and duplicates
        x = (np.round(x * 4) / 4).astype(np.float32)
    elif kind == "dup":        # 10 % exact duplicates
        k = max(n // 10, 1)
        x[:, RNG.choice(n, k, replace=False)] = x[:, RNG.choice(n, k, replace=False)]
    elif kind == "same":       # every point the same: every running minimum ties in every round (the selection's LDS atomic takes all 256 lanes' keys)
        x[:] = x[:, :1]
    elif kind == "padded":     # a quarter real points, the rest zero padding: the late rounds tie on a minimum of exactly 0
        x[:, n // 4:] = 0.0
    return x


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


FPS_CASES = [(2, 4096, 1024, "uniform"), (3, 1024, 128, "uniform"), (2, 128, 16, "uniform"), (2, 256, 64, "lattice"),
             (2, 64, 16, "uniform"), (1, 1000, 100, "dup"), (2, 100, 37, "uniform"), (1, 600, 600, "lattice"),
             (2, 2048, 64, "uniform"), (1, 5000, 50, "uniform"), (1, 8192, 33, "lattice"), (1, 9000, 20, "uniform"),
             (1, 1, 1, "uniform"), (2, 3, 3, "uniform"), (1, 4096, 1024, "lattice"), (1, 512, 300, "dup"),
             (2, 4096, 64, "same"), (1, 1024, 32, "same"), (2, 4096, 1100, "padded"), (1, 2048, 600, "padded"), (1, 6000, 1600, "padded")]


@pytest.mark.parametrize("b,n,m,kind", FPS_CASES)
def test_fps_bit_exact(P, b, n, m, kind):
    x = cloud(b, n, kind)
    ref = o.furthestsampling(x, m)
    got = P.furthestsampling(dev(x), m).cpu().numpy()
    assert got.dtype == np.int32 and np.array_equal(got, ref), f"first mismatch col {np.argmax((got != ref).any(0))}"


def test_fps_over_more_clouds_than_cus_is_the_per_cloud_result(P):
    """A first-level launch over more clouds than CUs (the look-ahead pipeline's groups: 16 x 32 = 512 clouds) drops its LDS reserve so that two sampling
    workgroups share a CU (csrc/fps.hip launch_reg): same kernel, same samples.  300 clouds of 4096 points against the oracle on a slice and against the
    32-cloud launches (which keep the reserve) on all of them."""
    x = cloud(300, 4096, "uniform")
    xd = dev(x)
    got = P.furthestsampling(xd, 1024)
    ref = o.furthestsampling(x[:6], 1024)
    assert np.array_equal(got[:6].cpu().numpy(), ref)
    parts = torch.cat([P.furthestsampling(xd[i:i + 32].contiguous(), 1024) for i in range(0, 300, 32)])
    assert torch.equal(got, parts)


def test_fps_leaves_temp_like_reference():
    """temp holds the final running minima after the call (sampling_cuda_kernel.cu:94-95 writes it back every round)."""
    from patchaugnet_amd import _lib
    x = cloud(2, 777, "uniform")
    xd = dev(x)
    temp = torch.full((2, 777), 1e10, device="cuda")
    idx = torch.empty((2, 50), dtype=torch.int32, device="cuda")
    _lib.call("pa_furthestsampling", 2, 777, 50, _lib.ptr(xd), _lib.ptr(temp), _lib.ptr(idx))
    sel = x[np.arange(2)[:, None], idx.cpu().numpy()[:, :-1]]          # all but the last sample were applied
    d = x[:, :, None, :] - sel[:, None, :, :]
    d = d * d
    exp = ((d[..., 0] + d[..., 1]) + d[..., 2]).min(2)
    assert np.array_equal(temp.cpu().numpy(), exp.astype(np.float32))


KNN_CASES = [(2, 4096, 1024, 40, "uniform"), (2, 4096, 512, 20, "uniform"), (3, 1024, 128, 20, "uniform"),
             (2, 128, 16, 20, "uniform"), (2, 256, 64, 20, "lattice"), (1, 1000, 77, 40, "dup"), (2, 30, 7, 40, "uniform"),
             (1, 100, 10, 64, "uniform"), (1, 100, 10, 1, "uniform"), (1, 300, 9, 100, "uniform"), (1, 50, 5, 200, "lattice"),
             (1, 9000, 16, 20, "uniform"), (1, 64, 64, 64, "lattice")]


@pytest.mark.parametrize("b,n,m,k,kind", KNN_CASES)
def test_knn_bit_exact(P, b, n, m, k, kind):
    x = cloud(b, n, kind)
    q = x[:, RNG.choice(n, m, replace=False)] if m <= n else cloud(b, m)
    ri, rd = o.knnquery(k, x, q)
    gi, gd = P.knnquery_with_dist(k, dev(x), dev(q))
    assert np.array_equal(gi.cpu().numpy(), ri)
    assert np.array_equal(gd.cpu().numpy().view(np.uint32), rd.view(np.uint32))
    assert np.array_equal(P.knnquery(k, dev(x), dev(q)).cpu().numpy(), ri)


LANE_CASES = ["uniform", "lattice", "dup", "planar", "clustered", "outside", "nonfinite", "line", "ragged"]


@pytest.mark.parametrize("kind", LANE_CASES)
@pytest.mark.parametrize("k", [16, 20, 32])
def test_knn_cell_grid_kernels_on_stress_shapes_bit_exact(P, kind, k):
    """The cell-grid neighbour searches (csrc/knn_quad.hip: four lanes per query, the default at 2048..4096 source points; csrc/knn.hip: the
    wave-per-query grid kernel behind pa_knn_quad_enable(0)) on the shapes that stress the grid walk: exact ties beyond the queue (lattice, duplicates -> direct-insertion path), a degenerate axis (planar, line),
    dense clusters with far outliers (many shells), queries outside the cloud's box (no early stop), non-finite points, ragged n."""
    import zlib
    rng = np.random.default_rng(zlib.crc32(f"{kind}-{k}".encode()))
    b, n, m = 2, 4096, 1024
    x = (rng.random((b, n, 3), dtype=np.float32) * 2 - 1).astype(np.float32)
    q = None
    if kind == "lattice":
        x = (np.round(x * 4) / 4).astype(np.float32)
    elif kind == "dup":
        x[:, rng.choice(n, 400, replace=False)] = x[:, rng.choice(n, 400, replace=False)]
    elif kind == "planar":
        x[..., 2] = 0.25
    elif kind == "line":
        x[..., 1] = -0.5
        x[..., 2] = 0.125
    elif kind == "clustered":
        x[:, : n - 40] = (x[:, : n - 40] * 0.02 + 0.7).astype(np.float32)       # 4056 points inside one cell, 40 spread over the box
    elif kind == "outside":
        q = (rng.random((b, m, 3), dtype=np.float32) * 6 - 3).astype(np.float32)
    elif kind == "nonfinite":
        x[0, 7] = np.inf
        x[1, 100, 1] = np.nan
        x[1, 2000] = -np.inf
    elif kind == "ragged":
        n, m = 3001, 301
        x = x[:, :n].copy()
    if q is None:
        q = x[:, rng.choice(n, m, replace=False)].copy()
        q[:, ::5] += (rng.random((b, len(range(0, m, 5)), 3), dtype=np.float32) * 0.01).astype(np.float32)   # queries that are not cloud points
    ri, rd = o.knnquery(k, x, q)
    from patchaugnet_amd import _lib
    lib = _lib.lib()
    gi2, gd2 = P.knnquery_with_dist(k, dev(x), dev(q))                      # the default: four lanes per query (knn_quad.hip)
    assert np.array_equal(gi2.cpu().numpy(), ri) and np.array_equal(gd2.cpu().numpy().view(np.uint32), rd.view(np.uint32))
    lib.pa_knn_quad_enable.argtypes, lib.pa_knn_quad_enable.restype = [__import__("ctypes").c_int], None
    lib.pa_knn_quad_enable(0)
    try:
        gi3, gd3 = P.knnquery_with_dist(k, dev(x), dev(q))                  # the wave-per-query grid kernel on the same stress shapes
        torch.cuda.synchronize()
    finally:
        lib.pa_knn_quad_enable(1)
    assert np.array_equal(gi3.cpu().numpy(), ri) and np.array_equal(gd3.cpu().numpy().view(np.uint32), rd.view(np.uint32))


def test_knn_non_finite_points(P):
    x = cloud(1, 200)
    x[0, 5] = np.inf
    x[0, 9, 1] = np.nan
    q = x[:, 20:40].copy()
    ri, rd = o.knnquery(20, x, q)
    gi, gd = P.knnquery_with_dist(20, dev(x), dev(q))
    assert np.array_equal(gi.cpu().numpy(), ri) and np.array_equal(gd.cpu().numpy().view(np.uint32), rd.view(np.uint32))


@pytest.mark.parametrize("kind", ["uniform", "lattice", "dup", "planar", "clustered", "outside", "faraway", "nonfinite", "line", "ragged", "bigknown"])
def test_three_nn_grid_kernel_bit_exact(P, kind):
    """The cell-grid 3-NN (csrc/three_nn_grid.hip: 512..4096 known points, >= 1024 queries) against the oracle on the shapes that stress
    its walk: ties (lattice, duplicates), a degenerate axis, one dense cluster + outliers (many shells), queries outside the known cloud's
    box (near and 1e4 away), non-finite points on both sides, ragged sizes; and the fused-weights form against the brute-force kernel."""
    import zlib
    from patchaugnet_amd import _lib
    rng = np.random.default_rng(zlib.crc32(kind.encode()))
    b, n, m = 2, 4096, 1024
    kn = (rng.random((b, m, 3), dtype=np.float32) * 2 - 1).astype(np.float32)
    u = (rng.random((b, n, 3), dtype=np.float32) * 2 - 1).astype(np.float32)
    if kind == "lattice":
        kn, u = (np.round(kn * 4) / 4).astype(np.float32), (np.round(u * 4) / 4).astype(np.float32)
    elif kind == "dup":
        kn[:, rng.choice(m, 200, replace=False)] = kn[:, rng.choice(m, 200, replace=False)]
        u[:, :m] = kn
    elif kind == "planar":
        kn[..., 2] = 0.25
    elif kind == "line":
        kn[..., 1] = -0.5
        kn[..., 2] = 0.125
    elif kind == "clustered":
        kn[:, : m - 20] = (kn[:, : m - 20] * 0.02 + 0.7).astype(np.float32)
    elif kind == "outside":
        u = (u * 3).astype(np.float32)
    elif kind == "faraway":
        u[:, ::3] = (u[:, ::3] * 1e4).astype(np.float32)
    elif kind == "nonfinite":
        kn[0, 7] = np.inf
        kn[1, 100, 1] = np.nan
        u[0, 5] = np.nan
        u[1, 9, 2] = -np.inf
    elif kind == "ragged":
        n, m = 3001, 777
        kn, u = kn[:, :m].copy(), u[:, :n].copy()
    elif kind == "bigknown":
        m = 4000
        kn = (rng.random((b, m, 3), dtype=np.float32) * 2 - 1).astype(np.float32)
    rd, ri = o.nearestneighbor(u, kn)
    gd, gi = P.nearestneighbor(dev(u), dev(kn))
    assert np.array_equal(gi.cpu().numpy(), ri)
    assert np.array_equal(gd.cpu().numpy(), np.sqrt(rd))
    lib = _lib.lib()
    lib.pa_three_nn_grid_enable.argtypes, lib.pa_three_nn_grid_enable.restype = [__import__("ctypes").c_int], None
    ud, kd = dev(u), dev(kn)
    w1, i1, w0, i0 = (torch.empty((b, n, 3), device="cuda"), torch.empty((b, n, 3), dtype=torch.int32, device="cuda"),
                      torch.empty((b, n, 3), device="cuda"), torch.empty((b, n, 3), dtype=torch.int32, device="cuda"))
    _lib.call("pa_three_nn_weights", b, n, m, _lib.ptr(ud), _lib.ptr(kd), _lib.ptr(w1), _lib.ptr(i1))
    lib.pa_three_nn_grid_enable(0)
    try:
        _lib.call("pa_three_nn_weights", b, n, m, _lib.ptr(ud), _lib.ptr(kd), _lib.ptr(w0), _lib.ptr(i0))
        torch.cuda.synchronize()
    finally:
        lib.pa_three_nn_grid_enable(1)
    assert torch.equal(i1, i0) and torch.equal(w1.view(torch.int32), w0.view(torch.int32))


NN_CASES = [(2, 4096, 1024, "uniform"), (2, 1024, 128, "uniform"), (2, 128, 16, "lattice"), (1, 50, 2, "uniform"),
            (1, 40, 1, "uniform"), (1, 300, 2500, "uniform"), (2, 257, 100, "dup")]


@pytest.mark.parametrize("b,n,m,kind", NN_CASES)
def test_three_nn_bit_exact(P, b, n, m, kind):
    u, kn = cloud(b, n, kind), cloud(b, m, kind)
    rd, ri = o.nearestneighbor(u, kn)
    gd, gi = P.nearestneighbor(dev(u), dev(kn))
    assert np.array_equal(gi.cpu().numpy(), ri)
    assert np.array_equal(gd.cpu().numpy(), np.sqrt(rd))          # wrapper returns sqrt(dist2), pointops.py:76


GROUP_CASES = [(2, 3, 4096, 1024, 20), (2, 64, 1024, 128, 20), (2, 256, 128, 16, 20), (1, 5, 100, 7, 3), (1, 1, 10, 1, 1),
               (2, 67, 333, 50, 20), (1, 2, 40000, 64, 4), (3, 16, 20000, 8, 5)]


@pytest.mark.parametrize("b,c,n,m,k", GROUP_CASES)
def test_grouping_and_gathering_bit_exact(P, b, c, n, m, k):
    f = RNG.standard_normal((b, c, n)).astype(np.float32)
    idx = RNG.integers(0, n, (b, m, k), dtype=np.int32)
    assert np.array_equal(P.grouping(dev(f), dev(idx)).cpu().numpy(), o.grouping_forward(f, idx))
    i1 = RNG.integers(0, n, (b, m), dtype=np.int32)
    assert np.array_equal(P.gathering(dev(f), dev(i1)).cpu().numpy(), o.gathering_forward(f, i1))
    assert np.array_equal(P.featuregather(dev(f), dev(i1)).cpu().numpy(), o.gathering_forward(f, i1))


def test_grouping_int(P):
    f = RNG.integers(-2**40, 2**40, (2, 3, 50), dtype=np.int64)
    idx = RNG.integers(0, 50, (2, 9, 4), dtype=np.int32)
    assert np.array_equal(P.grouping_int(dev(f), dev(idx)).cpu().numpy(), o.grouping_int_forward(f, idx))


@pytest.mark.parametrize("b,c,m,n", [(2, 512, 16, 128), (2, 256, 128, 1024), (2, 256, 1024, 4096), (1, 7, 33, 100), (1, 3, 20000, 50)])
def test_interpolation_bit_exact(P, b, c, m, n):
    f = RNG.standard_normal((b, c, m)).astype(np.float32)
    idx = RNG.integers(0, m, (b, n, 3), dtype=np.int32)
    w = RNG.random((b, n, 3), dtype=np.float32)
    w /= w.sum(-1, keepdims=True)
    got = P.interpolation(dev(f), dev(idx), dev(w)).cpu().numpy()
    assert np.array_equal(got, o.interpolation_forward(f, idx, w))


def test_backward_ops(P):
    b, c, n, m, k = 2, 6, 50, 12, 5
    f = torch.randn(b, c, n, device="cuda", requires_grad=True)
    idx = dev(RNG.integers(0, n, (b, m, k), dtype=np.int32))
    go = torch.randn(b, c, m, k, device="cuda")
    P.grouping(f, idx).backward(go)
    ref = o.grouping_backward(go.cpu().numpy(), idx.cpu().numpy(), n)
    assert np.allclose(f.grad.cpu().numpy(), ref, rtol=1e-5, atol=1e-5)
    f.grad = None
    i1 = dev(RNG.integers(0, n, (b, m), dtype=np.int32))
    g1 = torch.randn(b, c, m, device="cuda")
    P.gathering(f, i1).backward(g1)
    assert np.allclose(f.grad.cpu().numpy(), o.gathering_backward(g1.cpu().numpy(), i1.cpu().numpy(), n), rtol=1e-5, atol=1e-5)
    f.grad = None
    i3 = dev(RNG.integers(0, n, (b, 30, 3), dtype=np.int32))
    w = torch.rand(b, 30, 3, device="cuda")
    g3 = torch.randn(b, c, 30, device="cuda")
    P.interpolation(f, i3, w).backward(g3)
    ref = o.interpolation_backward(g3.cpu().numpy(), i3.cpu().numpy(), w.cpu().numpy(), n)
    assert np.allclose(f.grad.cpu().numpy(), ref, rtol=1e-5, atol=1e-5)


# (b, c, n source points, m centres, k): model shapes (LDS rows, one block per channel tile), a small b*c (source range split
# across workgroups + atomic merge), and n too large for the LDS rows (global-atomic kernel)
@pytest.mark.parametrize("b,c,n,m,k", [(3, 256, 1024, 4096, 20), (2, 64, 4096, 1024, 20), (1, 3, 4096, 1024, 20), (1, 5, 20000, 700, 4),
                                       (2, 19, 333, 77, 3)])
def test_backward_ops_at_scale(P, b, c, n, m, k):
    """K3 / K6 / K11 backward scatters vs the oracle (float sums in a different order: tolerance, not bit-exact)."""
    k = min(k, 8) if b * c * m * k > 4e7 else k
    f = torch.randn(b, c, n, device="cuda", requires_grad=True)
    idx = dev(RNG.integers(0, n, (b, m, k), dtype=np.int32))
    go = torch.randn(b, c, m, k, device="cuda")
    P.grouping(f, idx).backward(go)
    ref = o.grouping_backward(go.cpu().numpy(), idx.cpu().numpy(), n)
    assert np.allclose(f.grad.cpu().numpy(), ref, rtol=1e-4, atol=1e-4)
    f.grad = None
    i1 = dev(RNG.integers(0, n, (b, m), dtype=np.int32))
    g1 = torch.randn(b, c, m, device="cuda")
    P.gathering(f, i1).backward(g1)
    assert np.allclose(f.grad.cpu().numpy(), o.gathering_backward(g1.cpu().numpy(), i1.cpu().numpy(), n), rtol=1e-4, atol=1e-4)
    f.grad = None
    i3 = dev(RNG.integers(0, n, (b, m, 3), dtype=np.int32))          # m "unknown" points interpolated from the n known ones
    w = torch.rand(b, m, 3, device="cuda")
    g3 = torch.randn(b, c, m, device="cuda")
    P.interpolation(f, i3, w).backward(g3)
    ref = o.interpolation_backward(g3.cpu().numpy(), i3.cpu().numpy(), w.cpu().numpy(), n)
    assert np.allclose(f.grad.cpu().numpy(), ref, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("b,c,n,m", [(2, 256, 4096, 1024), (3, 18, 1023, 77), (1, 5, 1, 1), (2, 7, 2500, 8000), (1, 64, 4096, 16)])
def test_interpolation_backward_gather_form(b, c, n, m):
    """pa_interpolation_backward_gather (index list inverted once, plain sums) against the oracle; it ACCUMULATES into the caller's buffer like
    the reference's kernel, ragged sizes included; targets that nobody references keep their value."""
    from patchaugnet_amd import _lib
    g = np.random.default_rng(b * 1000 + n)
    go = g.standard_normal((b, c, n), dtype=np.float32)
    idx = g.integers(0, m, (b, n, 3), dtype=np.int32)
    if m > 4:
        idx[idx == 3] = 2                                     # target 3 is never referenced
    w = g.random((b, n, 3), dtype=np.float32)
    init = g.standard_normal((b, c, m), dtype=np.float32)
    out = dev(init.copy())
    scratch = torch.empty(_lib.lib().pa_interpolation_backward_scratch_ints(b, n, m), dtype=torch.int32, device="cuda")
    god, idxd, wd = dev(go), dev(idx), dev(w)                 # keep the device tensors alive across the asynchronous launch
    _lib.call("pa_interpolation_backward_gather", b, c, n, m, _lib.ptr(god), 0, _lib.ptr(idxd), _lib.ptr(wd), _lib.ptr(out), _lib.ptr(scratch))
    torch.cuda.synchronize()
    ref = init + o.interpolation_backward(go, idx, w, m)
    assert np.allclose(out.cpu().numpy(), ref, rtol=1e-4, atol=1e-4)
    if m > 4:
        assert np.array_equal(out.cpu().numpy()[:, :, 3], init[:, :, 3])


@pytest.mark.parametrize("n,m,r,k", [(4096, 1024, 0.2, 32), (200, 20, 0.5, 8), (3000, 300, 0.05, 16), (100, 10, 1e-4, 4)])
def test_ballquery_bit_exact(P, n, m, r, k):
    x = cloud(2, n)
    q = x[:, :m].copy()
    if r < 1e-3:
        q += 10
    assert np.array_equal(P.ballquery(r, k, dev(x), dev(q)).cpu().numpy(), o.ballquery(r, k, x, q))


def test_labelstat_and_featuredistribute(P):
    x, q = cloud(2, 300), cloud(2, 40)
    lab = RNG.integers(0, 5, (2, 300, 6), dtype=np.int32)
    r, k = 0.6, 12
    a, bidx = o.labelstat_and_ballquery(r, k, x, q, lab)
    ga, gb = P.labelstat_and_ballquery(r, k, dev(x), dev(q), dev(lab))
    assert np.array_equal(ga.cpu().numpy(), a) and np.array_equal(gb.cpu().numpy(), bidx)
    assert np.array_equal(P.labelstat_ballrange(r, dev(x), dev(q), dev(lab)).cpu().numpy(), o.labelstat_ballrange(r, x, q, lab))
    idx = RNG.integers(0, 300, (2, 40, 7), dtype=np.int32)
    assert np.array_equal(P.labelstat_idx(7, dev(lab), dev(idx)).cpu().numpy(), o.labelstat_idx(lab, idx))
    assert np.array_equal(P.featuredistribute(dev(x), dev(q)).cpu().numpy(), o.featuredistribute(x, q))


def test_rejects_cpu_tensors_and_bad_sizes(P):
    with pytest.raises(RuntimeError):
        P.furthestsampling(torch.rand(1, 64, 3), 8)
    from patchaugnet_amd import _lib
    x = torch.rand(1, 64, 3, device="cuda")
    with pytest.raises(RuntimeError):
        _lib.call("pa_knnquery", 1, 64, 0, 4, _lib.ptr(x), _lib.ptr(x), _lib.ptr(x), _lib.ptr(x))


def test_full_size_properties(P):
    """Config-2 sizes (B=32, 4096 points): properties that do not need the slow oracle at full batch."""
    from patchaugnet_amd.weights import synthetic_submaps
    x = synthetic_submaps(32, 4096, 1234).squeeze(1).cuda().contiguous()
    idx = P.furthestsampling(x, 1024)
    srt = torch.sort(idx.long(), dim=1)[0]
    assert (srt[:, 1:] != srt[:, :-1]).all() and (idx[:, 0] == 0).all()            # distinct samples, starts at 0
    ref = o.furthestsampling(x[:2].cpu().numpy(), 1024)
    assert np.array_equal(idx[:2].cpu().numpy(), ref)
    new_xyz = P.gathering(x.transpose(1, 2).contiguous(), idx).transpose(1, 2).contiguous()
    ki, kd = P.knnquery_with_dist(20, x, new_xyz)
    assert (ki[:, :, 0] == idx).all() and (kd[:, :, 0] == 0).all()                  # every centre is its own nearest
    assert (kd[:, :, 1:] >= kd[:, :, :-1]).all()                                     # sorted
    g = P.grouping(x.transpose(1, 2).contiguous(), ki)
    d = g - new_xyz.transpose(1, 2).unsqueeze(-1)
    assert torch.allclose((d * d).sum(1), kd, atol=1e-6)                             # gathered points are at dist2


@pytest.mark.parametrize("b,n,m,kind", [(2, 4096, 1024, "uniform"), (2, 1024, 128, "lattice"), (3, 128, 16, "uniform"), (1, 700, 50, "dup")])
def test_fps_gather_form(b, n, m, kind):
    """Engine form: no temp tensor, new_xyz written by the sampling kernel itself."""
    from patchaugnet_amd import _lib
    x = cloud(b, n, kind)
    xd = dev(x)
    idx = torch.empty((b, m), dtype=torch.int32, device="cuda")
    nx = torch.empty((b, m, 3), device="cuda")
    _lib.call("pa_furthestsampling_gather", b, n, m, _lib.ptr(xd), _lib.ptr(idx), _lib.ptr(nx))
    ref = o.furthestsampling(x, m)
    assert np.array_equal(idx.cpu().numpy(), ref)
    assert np.array_equal(nx.cpu().numpy(), np.take_along_axis(x, ref[..., None].astype(np.int64), 1))


@pytest.mark.parametrize("b,n,m", [(2, 4096, 1024), (2, 128, 16), (1, 77, 5)])
def test_three_nn_weights_form(b, n, m):
    from patchaugnet_amd import _lib
    u, kn = cloud(b, n), cloud(b, m)
    u[0, :3] = kn[0, :3]                      # exact coincidences: d = 0 -> 1/(0 + 1e-8)
    w = torch.empty((b, n, 3), device="cuda")
    idx = torch.empty((b, n, 3), dtype=torch.int32, device="cuda")
    ud, kd = dev(u), dev(kn)
    _lib.call("pa_three_nn_weights", b, n, m, _lib.ptr(ud), _lib.ptr(kd), _lib.ptr(w), _lib.ptr(idx))
    rd, ri = o.nearestneighbor(u, kn)
    assert np.array_equal(idx.cpu().numpy(), ri)
    r = 1.0 / (torch.sqrt(torch.from_numpy(rd)) + 1e-8)       # patch_aug_net.py:351-353
    ref = r / torch.sum(r, dim=2, keepdim=True)
    assert torch.allclose(w.cpu(), ref, rtol=1e-6, atol=1e-9)


def test_knnquery_naive_and_exclude_match_the_reference_definition():
    """pointops.py:367-404 / :436-473: first nsample columns (resp. columns 1..nsample) of the row-sorted distance matrix."""
    from patchaugnet_amd import pointops
    x = torch.rand(3, 500, 3, device="cuda")
    q = torch.rand(3, 70, 3, device="cuda")
    dist = (q.unsqueeze(2) - x.unsqueeze(1)).pow(2).sum(dim=3)
    order = torch.sort(dist, dim=2, stable=True)[1]
    assert torch.equal(pointops.knnquery_naive(9, x, q).long(), order[:, :, :9])
    assert torch.equal(pointops.knnquery_exclude(9, x, q).long(), order[:, :, 1:10])
    assert torch.equal(pointops.knnquery_exclude(4, x).long()[:, :, 0] != torch.arange(500, device="cuda"), torch.ones(3, 500, dtype=torch.bool, device="cuda"))


@pytest.mark.parametrize("b,n,m,parts", [(3, 4096, 1024, 4), (2, 1000, 250, 5), (1, 8192, 64, 2), (2, 300, 300, 3)])
def test_fps_in_ranges_equals_one_launch(b, n, m, parts):
    """pa_furthestsampling_range: the sampling order is prefix-stable (sampling_cuda_kernel.cu:59-168), so a chain of launches over ascending
    ranges -- running minima handed over through temp, last sample through idx -- gives the oracle's samples and coordinates bit for bit."""
    from patchaugnet_amd import _lib
    x = cloud(b, n, "uniform")
    xd = dev(x)
    ref = o.furthestsampling(x, m)
    idx = torch.full((b, m), -1, dtype=torch.int32, device="cuda")
    nx = torch.full((b, m, 3), float("nan"), device="cuda")
    temp = torch.empty((b, n), device="cuda")
    cuts = [round(m * i / parts) for i in range(parts + 1)]
    for j0, j1 in zip(cuts[:-1], cuts[1:]):
        _lib.call("pa_furthestsampling_range", b, n, m, j0, j1, _lib.ptr(xd), _lib.ptr(temp), _lib.ptr(idx), _lib.ptr(nx))
    assert np.array_equal(idx.cpu().numpy(), ref)
    assert np.array_equal(nx.cpu().numpy(), np.take_along_axis(x, ref[..., None].astype(np.int64), 1))


def test_knn_window_equals_the_rows_of_the_full_query(P):
    """pa_knnquery_window answers a window of every cloud's queries in place: same rows as the one-launch query, the rest untouched."""
    from patchaugnet_amd import _lib
    b, n, m, k = 3, 4096, 1024, 20
    x = cloud(b, n, "uniform")
    q = x[:, RNG.choice(n, m, replace=False)]
    gi, gd = P.knnquery_with_dist(k, dev(x), dev(q))
    idx = torch.full((b, m, k), -7, dtype=torch.int32, device="cuda")
    d2 = torch.full((b, m, k), -1.0, device="cuda")
    xd, qd = dev(x), dev(q)
    for q0 in (256, 768):
        _lib.call("pa_knnquery_window", b, n, m, k, q0, 256, _lib.ptr(xd), _lib.ptr(qd), _lib.ptr(idx), _lib.ptr(d2))
    for lo, hi, done in ((0, 256, False), (256, 512, True), (512, 768, False), (768, 1024, True)):
        if done:
            assert torch.equal(idx[:, lo:hi], gi[:, lo:hi]) and torch.equal(d2[:, lo:hi], gd[:, lo:hi])
        else:
            assert (idx[:, lo:hi] == -7).all() and (d2[:, lo:hi] == -1.0).all()


@pytest.mark.parametrize("kind", LANE_CASES)
def test_knn_presorted_cloud_equals_the_oracle(kind):
    """pa_cloud_cellsort + pa_knnquery_presorted (the cloud's counting sort as a launch of its own; the query workgroups copy the record) on the
    cell-grid stress shapes: exact ties, a degenerate axis, one dense cluster, queries outside the box, non-finite points, ragged n."""
    import zlib
    from patchaugnet_amd import _lib
    rng = np.random.default_rng(zlib.crc32(f"presort-{kind}".encode()))
    b, n, m, k = 2, 4096, 1024, 20
    x = (rng.random((b, n, 3), dtype=np.float32) * 2 - 1).astype(np.float32)
    q = None
    if kind == "lattice":
        x = (np.round(x * 4) / 4).astype(np.float32)
    elif kind == "dup":
        x[:, rng.choice(n, 400, replace=False)] = x[:, rng.choice(n, 400, replace=False)]
    elif kind == "planar":
        x[..., 2] = 0.25
    elif kind == "line":
        x[..., 1] = -0.5
        x[..., 2] = 0.125
    elif kind == "clustered":
        x[:, : n - 40] = (x[:, : n - 40] * 0.02 + 0.7).astype(np.float32)
    elif kind == "outside":
        q = (rng.random((b, m, 3), dtype=np.float32) * 6 - 3).astype(np.float32)
    elif kind == "nonfinite":
        x[0, 7] = np.inf
        x[1, 100, 1] = np.nan
        x[1, 2000] = -np.inf
    elif kind == "ragged":
        n, m = 3001, 301
        x = x[:, :n].copy()
    if q is None:
        q = x[:, rng.choice(n, m, replace=False)].copy()
    ri, rd = o.knnquery(k, x, q)
    xd, qd = dev(x), dev(q)
    cells = torch.empty(_lib.lib().pa_cloud_cellsort_floats(b, n), device="cuda")
    idx = torch.empty((b, m, k), dtype=torch.int32, device="cuda")
    d2 = torch.empty((b, m, k), device="cuda")
    _lib.call("pa_cloud_cellsort", b, n, _lib.ptr(xd), _lib.ptr(cells))
    _lib.call("pa_knnquery_presorted", b, n, m, k, _lib.ptr(xd), _lib.ptr(qd), _lib.ptr(cells), _lib.ptr(idx), _lib.ptr(d2))
    assert np.array_equal(idx.cpu().numpy(), ri)
    assert np.array_equal(d2.cpu().numpy().view(np.uint32), rd.view(np.uint32))


def test_pairwise_distances_device_path_and_autograd():
    """pointops.py:347-363.  Two fp32 device tensors run on the MFMA GEMM (norms + clamp in the epilogue) with AND without autograd (round 6: the
    gradient is two more launches of the same kernel, no torch.mm); strided views are accepted; gradients for x, for y, for the y = None form and
    with a non-uniform cotangent against float64 autograd of the reference's statement (the clamp's zero gradient included: duplicated rows)."""
    from patchaugnet_amd import pointops
    g = torch.Generator().manual_seed(3)
    x, y = torch.randn(300, 256, generator=g).cuda(), torch.randn(170, 256, generator=g).cuda()
    y[5] = x[7]                                                                   # an exact zero distance: the clamp decides its gradient
    ref = torch.clamp((x.double() ** 2).sum(1)[:, None] + (y.double() ** 2).sum(1)[None, :] - 2.0 * x.double() @ y.double().t(), min=0.0)
    with torch.no_grad():
        d = pointops.pairwise_distances(x, y)
        assert d.shape == (300, 170) and d.grad_fn is None
        assert torch.allclose(d.double(), ref, rtol=1e-5, atol=1e-3)
        ds = pointops.pairwise_distances(x.t().contiguous().t(), y[::2])          # a transposed (strided) view and a strided slice
        assert torch.allclose(ds.double(), ref[:, ::2], rtol=1e-5, atol=1e-3)
        dself = pointops.pairwise_distances(x)
        assert dself.shape == (300, 300) and float(dself.diagonal().abs().max()) < 1e-2
    cot = torch.randn(300, 170, generator=g).cuda()
    xg, yg = x.clone().requires_grad_(True), y.clone().requires_grad_(True)
    dg = pointops.pairwise_distances(xg, yg)
    assert dg.grad_fn is not None and "PairwiseDistances" in type(dg.grad_fn).__name__
    (dg * cot).sum().backward()
    xr, yr = x.double().clone().requires_grad_(True), y.double().clone().requires_grad_(True)
    dr = torch.clamp((xr ** 2).sum(1).view(-1, 1) + (yr ** 2).sum(1).view(1, -1) - 2.0 * torch.mm(xr, yr.t()), min=0.0)
    live = (dg.detach() > 0).double()                                             # the fp32 kernel's own clamp decisions (a distance of ~1e-5 may land on either side)
    (dr * cot.double() * live).sum().backward()
    assert torch.allclose(xg.grad.double(), xr.grad, rtol=1e-4, atol=2e-2) and torch.allclose(yg.grad.double(), yr.grad, rtol=1e-4, atol=2e-2)
    xs = x.clone().requires_grad_(True)                                           # y = None: both roles accumulate into x
    pointops.pairwise_distances(xs).sum().backward()
    gs = 4.0 * (x.double()[:, None, :] - x.double()[None, :, :]).sum(1)
    assert torch.allclose(xs.grad.double(), gs, rtol=1e-4, atol=5e-2)
    d2 = pointops.pairwise_distances(x.detach(), y.detach())                       # no gradient required: same kernel, same values
    assert d2.grad_fn is None and torch.equal(d2, dg.detach())
