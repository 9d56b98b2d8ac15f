"""Chamfer distance and KNN_CUDA-contract kNN on the MI355X against the oracle (bit-exact indices / distances)."""
import numpy as np
import pytest
import torch

from oracle import oracle_ops as o

pytestmark = pytest.mark.gpu
RNG = np.random.default_rng(5)


def pts(b, n, lattice=False):
    x = RNG.random((b, n, 3), dtype=np.float32)
    return (np.round(x * 8) / 8).astype(np.float32) if lattice else x


@pytest.mark.parametrize("B,n,m,lat", [(3072, 20, 20, False), (64, 20, 20, True), (3, 100, 333, False), (2, 700, 1300, False), (1, 4096, 4096, False)])
def test_chamfer_forward_bit_exact(B, n, m, lat):
    from patchaugnet_amd import chamfer_dist
    a, c = pts(B, n, lat), pts(B, m, lat)
    d1, d2, i1, i2 = o.chamfer_forward(a, c)
    g = chamfer_dist.forward(torch.from_numpy(a).cuda(), torch.from_numpy(c).cuda())
    assert np.array_equal(g[2].cpu().numpy(), i1) and np.array_equal(g[3].cpu().numpy(), i2)
    assert np.array_equal(g[0].cpu().numpy(), d1) and np.array_equal(g[1].cpu().numpy(), d2)


def test_chamfer_backward_and_l1_loss():
    from patchaugnet_amd import chamfer_dist
    a, c = pts(16, 20), pts(16, 20)
    ta = torch.from_numpy(a).cuda().requires_grad_(True)
    tc = torch.from_numpy(c).cuda().requires_grad_(True)
    loss = chamfer_dist.ChamferDistanceL1()(ta, tc)
    loss.backward()
    d1, d2, i1, i2 = o.chamfer_forward(a, c)
    ref_loss = (np.sqrt(d1).mean() + np.sqrt(d2).mean()) / 2                       # chamfer_dist/__init__.py:82-84
    assert abs(loss.item() - ref_loss) < 1e-6
    g1 = (0.5 / d1.size) * 0.5 / np.sqrt(d1)                                         # d loss / d dist1
    g2 = (0.5 / d2.size) * 0.5 / np.sqrt(d2)
    ra, rc = o.chamfer_backward(a, c, i1, i2, g1.astype(np.float32), g2.astype(np.float32))
    assert np.allclose(ta.grad.cpu().numpy(), ra, rtol=1e-4, atol=1e-7)
    assert np.allclose(tc.grad.cpu().numpy(), rc, rtol=1e-4, atol=1e-7)


@pytest.mark.parametrize("n,k,dim", [(10, 2, 5), (101, 10, 5), (1000, 10, 5), (1001, 400, 5), (3000, 26, 256), (500, 64, 256)])
def test_knn_generic_contract(n, k, dim):
    """Shapes of libs/KNN_CUDA/tests/test_knn_cuda.py:59-87 plus the retrieval shape (256-D, k = 26)."""
    from patchaugnet_amd import knn_cuda
    ref = RNG.random((n, dim), dtype=np.float32)
    qry = RNG.random((n // 2 + 1, dim), dtype=np.float32)
    rd, ri = o.knn_generic(ref.T.copy(), qry.T.copy(), k)
    d, i = knn_cuda.knn(torch.from_numpy(ref.T.copy()).cuda(), torch.from_numpy(qry.T.copy()).cuda(), k)
    assert i.dtype == torch.int64 and np.array_equal(i.cpu().numpy(), ri - 1)
    assert np.array_equal(d.cpu().numpy(), rd)
    D, I = knn_cuda.KNN(k, transpose_mode=True)(torch.from_numpy(ref[None]).cuda(), torch.from_numpy(qry[None]).cuda())
    assert D.shape == (1, len(qry), k) and np.array_equal(I[0].cpu().numpy(), (ri - 1).T)


@pytest.mark.parametrize("nref,nquery,k,dim", [(10, 10, 2, 5), (100, 51, 10, 5), (1000, 501, 10, 5), (1000, 1000, 400, 5), (3000, 1200, 26, 256)])
@pytest.mark.parametrize("batch", [1, 3])
def test_knn_module_against_kdtree(nref, nquery, k, dim, batch):
    """The reference's own native test (libs/KNN_CUDA/tests/test_knn_cuda.py:33-47, shapes :59-87 + the retrieval shape): distances of
    ``KNN(k, transpose_mode=True)`` equal ``sklearn.neighbors.KDTree.query`` to 3 decimals -- an answer neither this repo's oracle nor
    its kernels produced.  Indices: KDTree's distance at the returned index equals the returned distance (ties may order differently)."""
    from sklearn.neighbors import KDTree
    from patchaugnet_amd import knn_cuda
    rng = np.random.default_rng(nref * 7 + k)
    ref = rng.random((batch, nref, dim), dtype=np.float32)
    qry = rng.random((batch, nquery, dim), dtype=np.float32)
    D, I = knn_cuda.KNN(k, transpose_mode=True)(torch.from_numpy(ref).cuda(), torch.from_numpy(qry).cuda())
    assert D.shape == (batch, nquery, k) and I.shape == (batch, nquery, k) and I.dtype == torch.int64
    D, I = D.cpu().numpy(), I.cpu().numpy()
    for b in range(batch):
        td, ti = KDTree(ref[b].astype(np.float64)).query(qry[b].astype(np.float64), k=k)
        np.testing.assert_almost_equal(D[b], td, decimal=3)
        assert I[b].min() >= 0 and I[b].max() < nref
        direct = np.sqrt(((qry[b][:, None, :].astype(np.float64) - ref[b][I[b]].astype(np.float64)) ** 2).sum(-1))
        np.testing.assert_almost_equal(direct, td, decimal=3)
        assert all(len(set(r)) == k for r in I[b])                        # k distinct neighbours per query
    # non-transposed layout (knn_cuda/__init__.py:61-74): (bs, dim, n)
    D2, I2 = knn_cuda.KNN(k, transpose_mode=False)(torch.from_numpy(ref.transpose(0, 2, 1).copy()).cuda(),
                                                   torch.from_numpy(qry.transpose(0, 2, 1).copy()).cuda())
    assert np.array_equal(D2.cpu().numpy().transpose(0, 2, 1), D) and np.array_equal(I2.cpu().numpy().transpose(0, 2, 1), I)


EMD_FORMS = {"round_launches": 0, "one_workgroup": 1, "default": -1}


def _emd_form(form):
    """Select the auction's launch form: "default" / "round_launches" = one chip-wide launch per round; "one_workgroup" = the round-1 persistent kernel."""
    import ctypes
    from patchaugnet_amd import _lib
    lib = _lib.lib()
    lib.pa_emd_persistent_enable.argtypes, lib.pa_emd_persistent_enable.restype = [ctypes.c_int], None
    lib.pa_emd_persistent_enable(EMD_FORMS[form])


def _emd_gpu(a, c, eps, iters):
    """Run the HIP auction through the reference's module-level signature; returns status, dist, assignment and the state."""
    from patchaugnet_amd import emd_module
    x1, x2 = torch.from_numpy(a).cuda(), torch.from_numpy(c).cuda()
    b, n, _ = a.shape
    m = c.shape[1]
    z = lambda *s, dt=torch.float32: torch.zeros(*s, device="cuda", dtype=dt)
    st = {"dist": z(b, n), "assignment": z(b, n, dt=torch.int32) - 1, "price": z(b, m), "assignment_inv": z(b, m, dt=torch.int32) - 1,
          "bid": z(b, n, dt=torch.int32), "bid_increments": z(b, n), "max_increments": z(b, m), "max_idx": z(b * m, dt=torch.int32)}
    rc = emd_module.forward(x1, x2, st["dist"], st["assignment"], st["price"], st["assignment_inv"], st["bid"], st["bid_increments"],
                            st["max_increments"], None, None, None, None, st["max_idx"], eps, iters)
    torch.cuda.synchronize()
    return rc, {k: v.cpu().numpy() for k, v in st.items()}


@pytest.mark.parametrize("persistent", ["default", "one_workgroup"])
@pytest.mark.parametrize("b,n,eps,iters,lat", [(2, 1024, 0.005, 60, False), (3, 1024, 0.02, 1, False), (2, 2048, 0.01, 25, True),
                                               (1, 4096, 0.02, 12, False), (2, 1024, 0.002, 400, False), (20, 1024, 0.02, 30, False),
                                               (1, 8192, 0.02, 5, False), (16, 1024, 0.01, 3000, False)])
def test_emd_forward_matches_oracle_bit_exact(b, n, eps, iters, lat, persistent):
    """a-E: assignment, squared distances and the whole auction state equal the deterministic CPU restatement
    (emd_cuda.cu:228-282; free choices fixed as documented in oracle_emd_forward) -- in both launch forms: chip-wide (one launch per round,
    G workgroups per cloud, the round resolved by the last workgroup to arrive) and one persistent workgroup per cloud."""
    a, c = pts(b, n, lat), pts(b, n, lat)
    st, rd, ra, rs = o.emd_forward(a, c, eps, iters, full_state=True)
    _emd_form(persistent)
    try:
        rc, g = _emd_gpu(a, c, eps, iters)
    finally:
        _emd_form("default")
    assert rc == 1 and st == 1
    assert np.array_equal(g["assignment"], ra)
    assert np.array_equal(g["dist"], rd)
    assert np.array_equal(g["assignment_inv"], rs["assignment_inv"])
    assert np.array_equal(g["price"], rs["price"])
    assert np.array_equal(g["max_increments"], rs["max_increments"])


@pytest.mark.parametrize("eps,iters,expect_bijection", [(0.005, 300, False), (0.02, 3000, True), (0.005, 10000, True)])
def test_emd_cost_against_the_exact_assignment(eps, iters, expect_bijection):
    """a-E, held to something the oracle did not decide: the reference's auction races (emd_cuda.cu:181-215), so the oracle and the HIP kernel
    fix the free choices themselves and agree bit for bit -- this test asks whether that assignment is a good transport plan.
    With benefit a_ij = 3 - sqrt(d_ij) (emd_cuda.cu:142-155) an auction that keeps eps-complementary slackness ends within n * eps of the
    optimal total benefit; prices only rise and an object never loses its owner, so objects without an owner have price 0 and the bound
    sum_i sqrt(d_i) <= OPT + n * eps also covers the points the last round assigns without a contest (emd_cuda.cu:196-215: the result need
    not be a bijection, and a relaxed plan may even cost LESS than the optimal bijection).  OPT from scipy.optimize.linear_sum_assignment on
    the float64 distance matrix (CPU side of the test).  Checked for the oracle AND the HIP kernel:
      * upper bound (always):           mean sqrt(dist) <= OPT / n + eps
      * lower bounds:                   >= the mean nearest-neighbour distance (every plan), >= OPT / n when the plan is a bijection
      * (0.02, 3000) and (0.005, 10000) are long enough that the plan IS a bijection (asserted, so the two-sided bound is not vacuous);
        (0.005, 300) is the shape the loss runs at (pointnetvlad_loss.py:219-221), where a few points are still bidding."""
    from scipy.optimize import linear_sum_assignment
    rng = np.random.default_rng(11)
    a, c = rng.random((2, 1024, 3), dtype=np.float32), rng.random((2, 1024, 3), dtype=np.float32)
    n = a.shape[1]
    st, rd, ra, _ = o.emd_forward(a, c, eps, iters, full_state=True)
    rc, g = _emd_gpu(a, c, eps, iters)
    assert st == 1 and rc == 1
    slack = 1e-6                                               # fp32 sqrt / price arithmetic of the auction against the float64 statement
    for name, dist, ass in (("oracle", rd, ra), ("hip", g["dist"], g["assignment"])):
        for b in range(a.shape[0]):
            D = np.sqrt(((a[b][:, None, :].astype(np.float64) - c[b][None, :, :].astype(np.float64)) ** 2).sum(-1))
            assert ass[b].min() >= 0 and ass[b].max() < n
            # the reported squared distances are those of the reported assignment (emd_cuda.cu:217-226)
            assert np.allclose(np.sqrt(dist[b].astype(np.float64)), D[np.arange(n), ass[b]], rtol=1e-5, atol=1e-7)
            rows, cols = linear_sum_assignment(D)
            opt = D[rows, cols].sum() / n
            got = D[np.arange(n), ass[b]].sum() / n
            bij = len(np.unique(ass[b])) == n
            assert got <= opt + eps + slack, (name, b, got, opt)
            assert got >= D.min(axis=1).mean() - slack, (name, b)
            if expect_bijection:
                assert bij, (name, b, len(np.unique(ass[b])))
            if bij:
                assert got >= opt - slack, (name, b, got, opt)


def test_emd_shape_rules_and_properties():
    """emd_cuda.cu:236-249 return codes; identity clouds give zero cost; a permuted copy is recovered."""
    from patchaugnet_amd import emd_module
    a = pts(1, 1024)
    assert _emd_gpu(a[:, :1000], a[:, :1000], 0.02, 4)[0] == -1          # n % 1024 != 0 (the 20-point patch case)
    assert _emd_gpu(a, pts(1, 2048), 0.02, 4)[0] == -1                    # n != m
    perm = RNG.permutation(1024)
    x1 = torch.from_numpy(a).cuda()
    x2 = torch.from_numpy(a[:, perm].copy()).cuda()
    dist, ass = emd_module.emdModule()(x1, x2, 0.002, 300)
    assert float(dist.max()) == 0.0
    assert np.array_equal(perm[ass[0].cpu().numpy()], np.arange(1024))


def test_emd_backward_matches_oracle():
    from patchaugnet_amd import emd_module
    a, c = pts(2, 1024), pts(2, 1024)
    x1 = torch.from_numpy(a).cuda().requires_grad_(True)
    x2 = torch.from_numpy(c).cuda()
    dist, ass = emd_module.emdModule()(x1, x2, 0.01, 40)
    torch.sqrt(dist).mean(1).mean().backward()                             # pointnetvlad_loss.py:219-221
    d = dist.detach().cpu().numpy()
    g = (0.5 / np.sqrt(d) / d.size).astype(np.float32)
    ref = o.emd_backward(a, c, g, ass.cpu().numpy())
    assert np.allclose(x1.grad.cpu().numpy(), ref, rtol=1e-5, atol=1e-9)


def test_batched_hard_negative_mining_equals_per_query_path():
    """retrieval.get_hard_negatives_batch (one padded batch on the device) returns what the per-query kNN launches return."""
    from patchaugnet_amd import retrieval
    g = torch.Generator().manual_seed(4)
    ref = torch.nn.functional.normalize(torch.randn(3000, 256, generator=g), dim=1).cuda()
    rng = np.random.default_rng(1)
    negs = [rng.choice(3000, int(rng.integers(5, 700)), replace=False).tolist() for _ in range(150)]
    negs[3] = negs[3][:4]                                                    # fewer negatives than requested -> []
    qs = ref[:150] + 0.01
    got = retrieval.get_hard_negatives_batch(qs, ref, negs, num_hard_neg=10, chunk=64)
    exp = [retrieval.get_hard_negatives(q, ref, n, 10) for q, n in zip(qs, negs)]
    assert got == exp and got[3] == []


@pytest.mark.parametrize("i", [0, 1, 2, 3])
def test_fused_tuple_losses_match_the_reference_run(i):
    """triplet_loss / quadruplet_loss as one HIP launch (csrc/losses.hip: value + gradient) against the values and gradients the REFERENCE's
    losses/pointnetvlad_loss.py produced for the same inputs in float64 (tests/golden/losses.npz, oracle/gen_loss_golden.py)."""
    from oracle.gen_loss_golden import LOSS_CASES, loss_inputs
    from patchaugnet_amd import losses
    from tests._util import golden
    g = golden("losses")
    name, kw = LOSS_CASES[i]
    assert not kw.get("soft_margin", False)
    q, pos, neg, other = [t.float().cuda().requires_grad_(True) for t in loss_inputs()]
    fn = getattr(losses, name)
    v = fn(q, pos, neg, 0.5, **kw) if name == "triplet_loss" else fn(q, pos, neg, other, 0.5, 0.2, **kw)
    assert v.grad_fn is not None and "TupleLossFused" in type(v.grad_fn).__name__, "the fused kernel did not take the call"
    (v * 3.0).backward()                                        # a non-trivial upstream gradient
    assert abs(float(v) - float(g[f"loss{i}_value"])) <= 2e-6 * max(1.0, abs(float(g[f"loss{i}_value"])))
    for t, x in zip("qpno", (q, pos, neg, other)):
        got = x.grad.cpu().numpy() / 3.0 if x.grad is not None else np.zeros_like(g[f"loss{i}_grad_{t}"])
        assert np.allclose(got, g[f"loss{i}_grad_{t}"], rtol=2e-5, atol=2e-7), (name, t, np.abs(got - g[f"loss{i}_grad_{t}"]).max())


def test_fused_quadruplet_loss_on_the_training_tuple_shape():
    """One query, 2 positives, 14 negatives (configs/patch_aug_net.yaml:60-62), inactive and active hinges, against the torch statement in fp64."""
    from patchaugnet_amd import losses
    g = torch.Generator().manual_seed(2)
    for scale in (0.05, 1.0):                       # 0.05: every negative far beyond the margin -> zero loss, zero gradient
        q = torch.nn.functional.normalize(torch.randn(1, 1, 256, generator=g), dim=-1)
        pos = torch.nn.functional.normalize(q + scale * 0.5 * torch.randn(1, 2, 256, generator=g), dim=-1)
        neg = torch.nn.functional.normalize(q + (1.2 if scale == 1.0 else 30.0) * torch.randn(1, 14, 256, generator=g), dim=-1)
        oth = torch.nn.functional.normalize(torch.randn(1, 1, 256, generator=g), dim=-1)
        ref_in = [t.double().requires_grad_(True) for t in (q, pos, neg, oth)]
        ref = losses.quadruplet_loss(*ref_in, 0.5, 0.2, use_min=False, lazy=True)
        ref.backward()
        dev_in = [t.cuda().requires_grad_(True) for t in (q, pos, neg, oth)]
        got = losses.quadruplet_loss(*dev_in, 0.5, 0.2, use_min=False, lazy=True)
        got.backward()
        assert abs(float(got) - float(ref)) <= 2e-6
        for a, b in zip(dev_in, ref_in):
            assert torch.allclose(a.grad.cpu().double(), b.grad if b.grad is not None else torch.zeros_like(b), rtol=2e-5, atol=2e-7)
