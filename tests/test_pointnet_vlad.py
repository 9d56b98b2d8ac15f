"""BASELINE.json configs[0] -- PointNetVLAD (dense torch ops only in the reference): the re-declared class against vectors
from the reference's own class (tests/golden/pointnet_vlad.npz, oracle/gen_golden.py), state-dict key parity, batch = 1."""
import numpy as np
import pytest
import torch

from tests._util import golden, key_table, seeded_sd_from_table


def _model(npts, device="cpu"):
    from patchaugnet_amd.pointnet_vlad import PointNetVlad
    m = PointNetVlad(global_feat=True, feature_transform=True, max_pool=False, output_dim=256, num_points=npts)   # evaluate.py:88-90
    m.load_state_dict(seeded_sd_from_table("pointnet_vlad"), strict=True)
    return m.to(device).eval()


def test_state_dict_keys_equal_the_reference():
    from patchaugnet_amd.pointnet_vlad import PointNetVlad
    m = PointNetVlad(global_feat=True, feature_transform=True, max_pool=False, output_dim=256, num_points=4096)
    tab = key_table("pointnet_vlad")
    assert list(m.state_dict().keys()) == list(tab.keys())
    assert all(list(v.shape) == tab[k][0] for k, v in m.state_dict().items())
    assert sum(p.numel() for p in m.parameters()) == 19779145


@pytest.mark.parametrize("tag,npts", [("small", 512), ("full", 4096)])
def test_cpu_forward_matches_reference_vectors(tag, npts):
    g = golden("pointnet_vlad")
    m = _model(npts)
    x = torch.from_numpy(g[f"{tag}_x"])
    with torch.no_grad():
        d = m(x)
        d1 = m(x[:1])                                                          # batch = 1 (configs[0])
    assert np.abs(d.numpy() - g[f"{tag}_desc"]).max() <= 2e-5
    assert np.abs(d1.numpy() - g[f"{tag}_desc"][:1]).max() <= 2e-5


@pytest.mark.gpu
@pytest.mark.parametrize("tag,npts", [("small", 512), ("full", 4096)])
def test_gpu_forward_matches_reference_vectors(tag, npts):
    """evaluate.py:88-104 moves the model to the accelerator: the device form (train_ops: hand-written MFMA GEMM / BatchNorm / NetVLAD
    kernels, channel-major) against the vectors of the reference's own class.  fp32 GEMMs in another summation order: 1e-4 on the
    (un-normalised, gated) 256-D output, the tolerance of the PatchAugNet descriptors."""
    g = golden("pointnet_vlad")
    m = _model(npts, "cuda")
    x = torch.from_numpy(g[f"{tag}_x"]).cuda()
    with torch.no_grad():
        d = m(x)
        d1 = m(x[:1])
    ref = g[f"{tag}_desc"]
    assert np.abs(d.cpu().numpy() - ref).max() <= 1e-4 * max(1.0, np.abs(ref).max())
    assert np.abs(d1.cpu().numpy() - ref[:1]).max() <= 1e-4 * max(1.0, np.abs(ref).max())


@pytest.mark.gpu
def test_gpu_path_reaches_no_library_gemm():
    """The device form runs on this package's kernels only: no rocBLAS / hipBLASLt / MIOpen kernel in a forward + backward."""
    from torch.profiler import ProfilerActivity, profile
    m = _model(512, "cuda").train()
    x = torch.rand(2, 1, 512, 3, device="cuda") * 2 - 1
    m(x).sum().backward()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        m(x).sum().backward()
        torch.cuda.synchronize()
    names = [e.key for e in prof.key_averages()]
    bad = [n for n in names if any(t in n.lower() for t in ("cijk", "rocblas", "hipblas", "miopen", "gemm_kernel", "tensile"))]
    assert not bad, bad
    assert any("tgemm" in n for n in names), names[:10]
