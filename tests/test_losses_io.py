"""Training losses and the submap loader / normaliser against results from the reference's own functions
(tests/golden/losses.npz, made by oracle/gen_loss_golden.py from losses/pointnetvlad_loss.py and utils/loading_pointclouds.py)."""
import numpy as np
import pytest
import torch

from oracle.gen_loss_golden import LOSS_CASES, call_loss, cloud_inputs, loss_inputs
from tests._util import golden


@pytest.mark.parametrize("i", range(len(LOSS_CASES)))
def test_descriptor_losses_match_reference(i):
    from patchaugnet_amd import losses
    g = golden("losses")
    name, kw = LOSS_CASES[i]
    v, grads = call_loss(getattr(losses, name), name, kw, *loss_inputs())
    assert abs(float(v) - float(g[f"loss{i}_value"])) <= 1e-12
    for t, gr in zip("qpno", grads):
        assert np.allclose(gr, g[f"loss{i}_grad_{t}"], rtol=1e-5, atol=1e-8), (name, t)


def test_contrastive_patch_loss_matches_reference():
    from patchaugnet_amd import losses
    g = golden("losses")
    gen = torch.Generator().manual_seed(11)
    lists = [[torch.randn(256, generator=gen, dtype=torch.float64) for _ in range(9)] for _ in range(3)]
    assert abs(float(losses.contrastive_loss(lists[0], lists[1], lists[2], 0.5)) - float(g["contrastive_value"])) <= 1e-12
    assert abs(float(losses.contrastive_loss(lists[0], [], lists[2], 25.0)) - float(g["contrastive_value_nopos"])) <= 1e-12
    assert losses.get_loss_func("quadruplet") is losses.quadruplet_loss
    assert losses.get_loss_func("anything else") is losses.triplet_loss_wrapper      # train_place_recognition.py:116-117


def test_loader_and_normalisation_match_reference(tmp_path):
    from patchaugnet_amd import io
    g = golden("losses")
    pc = cloud_inputs()
    path = tmp_path / "cloud.bin"
    pc.astype(np.float64).tofile(path)                                               # Oxford .bin: raw float64 triples
    loaded = io.load_pc_file(str(path))
    assert loaded.dtype == np.float64 and np.array_equal(loaded, pc)
    n1, meta = io.normalize_point_cloud(loaded, return_norm_meta=True)
    assert np.array_equal(n1, g["norm_zoom"]) and meta["scale"] == float(g["norm_scale"]) and np.array_equal(meta["trans"], g["norm_trans"])
    assert np.array_equal(io.normalize_point_cloud(loaded, zoom=False), g["norm_nozoom"])
    assert np.abs(np.linalg.norm(n1, axis=1).max() - 1.0) < 1e-12
    assert io.load_pc_files(["cloud.bin", "missing.bin"], str(tmp_path))[0].shape == (777, 3)
    with pytest.raises(NotImplementedError):
        io.load_pc_file(str(path), input_dim=13)


def test_batch_stager_cpu_roundtrip():
    from patchaugnet_amd import io
    st = io.BatchStager(4, num_points=16, device="cpu", depth=2)
    rs = np.random.RandomState(0)
    for _ in range(5):                                                               # more batches than ring slots
        clouds = [rs.standard_normal((16, 3)) for _ in range(3)]
        x = st.stage(clouds)
        assert x.shape == (3, 1, 16, 3) and x.dtype == torch.float32
        assert np.array_equal(x[:, 0].numpy(), np.stack(clouds).astype(np.float32))


@pytest.mark.gpu
def test_batch_stager_and_point_set_losses_on_gpu():
    from oracle import oracle_ops as o
    from patchaugnet_amd import io, losses
    st = io.BatchStager(8, num_points=64, device="cuda")
    rs = np.random.RandomState(1)
    clouds = [rs.standard_normal((64, 3)) for _ in range(8)]
    x = st.stage(clouds)
    torch.cuda.synchronize()
    assert np.array_equal(x[:, 0].cpu().numpy(), np.stack(clouds).astype(np.float32))
    a = [torch.rand(64, 20, 3, device="cuda") for _ in range(2)]
    b = [torch.rand(64, 20, 3, device="cuda") for _ in range(2)]
    v = losses.patch_chamfer_loss(a, b)
    d1, d2, _, _ = o.chamfer_forward(torch.cat(a).cpu().numpy(), torch.cat(b).cpu().numpy())
    assert abs(float(v) - (np.sqrt(d1).mean() + np.sqrt(d2).mean()) / 2) < 1e-6
    with pytest.raises(ValueError):
        losses.patch_emd_loss(a, b)                                                  # 20-point patches: SURVEY.md section 9.3
    p1, p2 = [torch.rand(1, 4096, 3, device="cuda")], [torch.rand(1, 4096, 3, device="cuda")]
    e = losses.emd_loss(p1, p2)
    st_, dist, _ = o.emd_forward(p1[0].cpu().numpy(), p2[0].cpu().numpy(), 0.02, 1024)
    assert st_ == 1 and abs(float(e) - np.sqrt(dist).mean()) < 1e-6


def test_descriptor_cache_files_are_the_references(tmp_path):
    """make_descs(save=True) (scene_dataset.py:689-707) and get_g_desc / get_l_kpt_desc (:784-798, :807-831): file names, pickle
    protocol, shapes and dtypes -- the global file is byte-identical to what the reference's three lines write."""
    import pickle
    from patchaugnet_amd import io as pio
    rs = np.random.RandomState(0)
    B, N, K, C = 3, 64, 16, 8
    feed = torch.from_numpy(rs.standard_normal((B, 1, N, 3)).astype(np.float32))
    g = torch.from_numpy(rs.standard_normal((B, C)).astype(np.float32))
    fp = [torch.zeros(B, C, 4, 1), torch.from_numpy(rs.standard_normal((B, C, K, 1)).astype(np.float32)), torch.zeros(B, C, N, 1)]
    ci = [torch.from_numpy(np.stack([rs.permutation(N)[:K] for _ in range(B)]).astype(np.int32))]
    metas = [{"scale": 2.0 + i, "trans": np.array([1.0, 2.0, 3.0]) * i} for i in range(B)]
    gd, ld = str(tmp_path / "g"), str(tmp_path / "l")
    pio.save_descriptor_cache(gd, ld, 40, g, feed, fp, ci, metas)
    for b in range(B):
        want = pickle.dumps(g.numpy()[b].reshape(1, -1), protocol=pickle.HIGHEST_PROTOCOL)
        assert open(f"{gd}/{40 + b}.pickle", "rb").read() == want
        got = pio.load_global_descriptor(gd, 40 + b)
        assert got.shape == (1, C) and got.dtype == np.float32 and np.array_equal(got[0], g.numpy()[b])
        kpt, desc, meta = pio.load_local_descriptor(ld, 40 + b)
        assert kpt.dtype == np.float64 and np.array_equal(kpt, feed.numpy()[b, 0][ci[0].numpy()[b]].astype(np.float64))
        assert np.array_equal(desc, fp[1].numpy()[b, :, :, 0].T) and meta["scale"] == 2.0 + b
        world, _, _ = pio.load_local_descriptor(ld, 40 + b, unify_coord=True, global_offset=np.array([0.5, 0.5, 0.5]))
        assert np.allclose(world, kpt * (2.0 + b) + (metas[b]["trans"].reshape(1, 3) - 0.5))
    assert pio.load_global_descriptor(gd, 7) is None and pio.load_local_descriptor(ld, 7) is None
    assert pio.load_global_descriptors(gd, [41, 40]).shape == (2, C)


def test_hard_negative_refresh_schedule():
    """train_place_recognition.py:401-406: every 1400 // batch_size batches at phase 29, only past hard_neg_epoch."""
    from patchaugnet_amd.train import hard_negative_refresh_due as due
    assert [c for c in range(1, 800) if due(c, 4, 6, 5)] == [29, 379, 729]
    assert not any(due(c, 4, 5, 5) for c in range(1, 800))
    assert not any(due(c, 4, 9, 5, use_hard_neg=False) for c in range(1, 800))
