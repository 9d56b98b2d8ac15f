"""The CPU oracle (oracle/pointops_oracle.c) against independent numpy statements
of the same semantics.  No GPU needed."""
import numpy as np
import pytest

from oracle import oracle_ops as o

RNG = np.random.default_rng(7)


def cloud(b, n, dup=False):
    x = (RNG.random((b, n, 3), dtype=np.float32) * 2 - 1)
    if dup:  # exact duplicates and a coarse lattice => many exact distance ties
        x = np.round(x * 4) / 4
    return x.astype(np.float32)


def d2_matrix(q, p):
    """((dx*dx + dy*dy) + dz*dz) in fp32, q (m,3) p (n,3) -> (m,n)."""
    d = q[:, None, :] - p[None, :, :]
    s = d * d
    return (s[..., 0] + s[..., 1]) + s[..., 2]


def bitrev(x, bits):
    r = 0
    for i in range(bits):
        r = (r << 1) | ((x >> i) & 1)
    return r


def fps_numpy(p, m):
    """Appendix A.1: argmax under (temp desc, bitrev(k mod bs) asc, k asc)."""
    n = len(p)
    bs = o.opt_n_threads(n)
    bits = int(np.log2(bs))
    rank = np.array([bitrev(k % bs, bits) * (n + 1) + k for k in range(n)])
    temp = np.full(n, 1e10, np.float32)
    out, old = [0], 0
    for _ in range(1, m):
        d = p - p[old]
        d = d * d
        temp = np.minimum(((d[:, 0] + d[:, 1]) + d[:, 2]).astype(np.float32), temp)
        cand = np.where(temp == temp.max())[0]
        old = int(cand[np.argmin(rank[cand])])
        out.append(old)
    return np.array(out, np.int32)


@pytest.mark.parametrize("n", [1, 2, 3, 16, 100, 128, 1000, 1024, 4096, 5000])
def test_opt_n_threads(n):
    t = o.opt_n_threads(n)
    assert t == min(1 << (n.bit_length() - 1), 1024)


@pytest.mark.parametrize("n,m,dup", [(64, 16, False), (100, 37, False), (128, 16, True), (1024, 128, False),
                                     (4096, 256, False), (600, 600, True)])
def test_fps_total_order(n, m, dup):
    x = cloud(2, n, dup)
    idx = o.furthestsampling(x, m)
    for b in range(2):
        assert np.array_equal(idx[b], fps_numpy(x[b], m))


def test_fps_all_identical_points():
    x = np.zeros((1, 64, 3), np.float32)
    idx = o.furthestsampling(x, 8)
    assert idx[0, 0] == 0 and np.all(idx[0, 1:] == 0)   # every distance 0 > -1 -> lowest bit-reversed tid = 0


@pytest.mark.parametrize("n,m,k,dup", [(256, 32, 20, False), (1024, 128, 40, False), (128, 16, 20, True), (30, 7, 40, False)])
def test_knn_stable_order(n, m, k, dup):
    x = cloud(2, n, dup)
    q = x[:, RNG.choice(n, m, replace=False)]
    idx, d2 = o.knnquery(k, x, q)
    for b in range(2):
        dm = d2_matrix(q[b], x[b])
        order = np.argsort(dm, axis=1, kind="stable")
        kk = min(k, n)
        assert np.array_equal(idx[b, :, :kk], order[:, :kk])
        assert np.array_equal(d2[b, :, :kk], np.take_along_axis(dm, order[:, :kk], 1))
        if k > n:   # unfilled slots: idx 0, dist +inf (knnquery_cuda_kernel.cu:23-26, :44-47)
            assert np.all(idx[b, :, n:] == 0) and np.all(np.isinf(d2[b, :, n:]))


@pytest.mark.parametrize("n,m,dup", [(512, 64, False), (128, 16, True), (50, 2, False), (40, 1, False)])
def test_three_nn(n, m, dup):
    u, kn = cloud(2, n, dup), cloud(2, m, dup)
    d2, idx = o.nearestneighbor(u, kn)
    for b in range(2):
        dm = d2_matrix(u[b], kn[b])
        order = np.argsort(dm, axis=1, kind="stable")[:, :3]
        kk = min(3, m)
        assert np.array_equal(idx[b, :, :kk], order[:, :kk])
        assert np.array_equal(d2[b, :, :kk], np.take_along_axis(dm, order[:, :kk], 1))
        if m < 3:
            assert np.all(idx[b, :, m:] == 0) and np.all(np.isinf(d2[b, :, m:]))


def test_gather_group_interpolate():
    b, c, n, m, k = 2, 5, 64, 16, 4
    f = RNG.random((b, c, n), dtype=np.float32)
    i1 = RNG.integers(0, n, (b, m), dtype=np.int32)
    i2 = RNG.integers(0, n, (b, m, k), dtype=np.int32)
    assert np.array_equal(o.gathering_forward(f, i1), np.take_along_axis(f, i1[:, None, :].repeat(c, 1).astype(np.int64), 2))
    g = o.grouping_forward(f, i2)
    for bi in range(b):
        assert np.array_equal(g[bi], f[bi][:, i2[bi]])
    i3 = RNG.integers(0, n, (b, m, 3), dtype=np.int32)
    w = RNG.random((b, m, 3), dtype=np.float32)
    out = o.interpolation_forward(f, i3, w)
    for bi in range(b):
        p = f[bi][:, i3[bi]]                       # (c, m, 3)
        ref = (w[bi][None, :, 0] * p[..., 0] + w[bi][None, :, 1] * p[..., 1]) + w[bi][None, :, 2] * p[..., 2]
        assert np.array_equal(out[bi], ref.astype(np.float32))
    # backward ops are transposes of the forward gathers
    go = RNG.random((b, c, m, k), dtype=np.float32)
    gb = o.grouping_backward(go, i2, n)
    ref = np.zeros((b, c, n), np.float64)
    for bi in range(b):
        for j in range(m):
            for s in range(k):
                ref[bi, :, i2[bi, j, s]] += go[bi, :, j, s]
    assert np.allclose(gb, ref, rtol=1e-5, atol=1e-6)


def test_ballquery():
    x = cloud(2, 200)
    q = x[:, :20]
    r, k = 0.5, 8
    idx = o.ballquery(r, k, x, q)
    r2 = np.float32(r) * np.float32(r)
    for b in range(2):
        dm = d2_matrix(q[b], x[b])
        for j in range(20):
            hits = np.where(dm[j] < r2)[0][:k]
            exp = np.zeros(k, np.int32)
            if len(hits):
                exp[:] = hits[0]
                exp[:len(hits)] = hits
            assert np.array_equal(idx[b, j], exp)
    far = o.ballquery(1e-4, k, x, q + 10)      # no hit: slots keep the caller's zeros
    assert np.all(far == 0)


def test_chamfer_forward_backward():
    a, c = cloud(3, 20), cloud(3, 33)
    d1, d2, i1, i2 = o.chamfer_forward(a, c)
    for b in range(3):
        dm = d2_matrix(a[b], c[b])    # (buf - x1) squared is symmetric in sign
        assert np.array_equal(i1[b], dm.argmin(1)) and np.array_equal(d1[b], dm.min(1))
        assert np.array_equal(i2[b], dm.argmin(0)) and np.array_equal(d2[b], dm.min(0))
    big_a, big_c = cloud(1, 700), cloud(1, 1300)     # more than one 512-point tile
    D1, D2, I1, I2 = o.chamfer_forward(big_a, big_c)
    dm = d2_matrix(big_a[0], big_c[0])
    assert np.array_equal(I1[0], dm.argmin(1)) and np.array_equal(I2[0], dm.argmin(0))
    g1, g2 = RNG.random(d1.shape, dtype=np.float32), RNG.random(d2.shape, dtype=np.float32)
    ga, gc = o.chamfer_backward(a, c, i1, i2, g1, g2)
    ra, rc = np.zeros(a.shape, np.float64), np.zeros(c.shape, np.float64)
    for b in range(3):
        for j in range(20):
            v = 2 * g1[b, j] * (a[b, j] - c[b, i1[b, j]]); ra[b, j] += v; rc[b, i1[b, j]] -= v
        for j in range(33):
            v = 2 * g2[b, j] * (c[b, j] - a[b, i2[b, j]]); rc[b, j] += v; ra[b, i2[b, j]] -= v
    assert np.allclose(ga, ra, atol=1e-5) and np.allclose(gc, rc, atol=1e-5)


@pytest.mark.parametrize("n,k", [(10, 2), (101, 10), (1000, 10), (1001, 400)])
def test_knn_generic_vs_kdtree(n, k):
    """Shapes and tolerance of the reference's only native test, libs/KNN_CUDA/tests/test_knn_cuda.py:59-87
    (distances vs sklearn KDTree, decimal=3); indices are additionally pinned by a stable argsort."""
    from sklearn.neighbors import KDTree
    dim = 5
    ref = RNG.random((n, dim), dtype=np.float32)
    qry = RNG.random((n // 2 + 1, dim), dtype=np.float32)
    dist, ind = o.knn_generic(ref.T.copy(), qry.T.copy(), k)
    kd, _ = KDTree(ref).query(qry, k=k)
    np.testing.assert_almost_equal(dist.T, kd, decimal=3)
    ssd = np.zeros((len(qry), n), np.float32)
    for d in range(dim):
        t = ref[None, :, d] - qry[:, None, d]
        ssd = ssd + t * t
    order = np.argsort(ssd, axis=1, kind="stable")[:, :k]
    assert np.array_equal(ind.T - 1, order)


def test_emd_properties():
    """Auction EMD (emd_cuda.cu:228-282): shape rules, identity, permutation recovery."""
    x = RNG.random((2, 1024, 3), dtype=np.float32)
    assert o.emd_forward(x[:, :1000], x[:, :1000], 0.02, 4)[0] == -1       # n % 1024 != 0 -> -1
    st, dist, ass = o.emd_forward(x, x, 0.005, 50)
    assert st == 1 and np.all(dist < 1e-3)
    perm = RNG.permutation(1024)
    st, dist, ass = o.emd_forward(x, x[:, perm], 0.002, 300)
    assert np.mean(np.sqrt(dist)) < 0.01
    assert len(np.unique(ass[0])) > 1000                                   # near-bijection
    y = RNG.random((2, 1024, 3), dtype=np.float32)
    st, dist, ass = o.emd_forward(x, y, 0.005, 200)
    nn = np.sqrt(d2_matrix(x[0], y[0]).min(1)).mean()
    assert np.sqrt(dist[0]).mean() >= nn - 1e-6                            # EMD >= one-sided NN cost
