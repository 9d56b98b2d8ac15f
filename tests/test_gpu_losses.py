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
