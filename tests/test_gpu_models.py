"""End-to-end parity on the MI355X: the product model classes (HIP ops; module path and fused engine) against
(a) vectors produced by the reference's own Python classes (tests/golden) and (b) the CPU oracle on fresh inputs.
Tolerances: indices bit-exact; L2-normalised descriptors max|d| <= 1e-4 and cosine >= 0.99999."""
import numpy as np
import pytest
import torch

from patchaugnet_amd import configs
from tests._util import golden, seeded_sd_from_table, samples, summarize

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _pan(cfg):
    from patchaugnet_amd import patch_aug_net
    m = patch_aug_net.Network(param=cfg, use_a2a_recon=True, use_l2_norm=True)
    m.load_state_dict(seeded_sd_from_table("patch_aug_net"), strict=True)
    return m.cuda().eval()


def _check_desc(d, ref):
    d = d.detach().cpu().numpy()
    assert np.abs(d - ref).max() <= TOL, np.abs(d - ref).max()
    cos = (d * ref).sum(1) / (np.linalg.norm(d, axis=1) * np.linalg.norm(ref, axis=1))
    assert cos.min() >= 0.99999, cos.min()


@pytest.mark.parametrize("tag", ["small", "full"])
@pytest.mark.parametrize("fused", [False, True])
def test_patch_aug_net_vs_reference_vectors(tag, fused):
    g = golden("patch_aug_net")
    cfg = configs.patch_aug_net_config()
    if tag == "small":
        cfg = configs.scaled_config(cfg, 512)
    m = _pan(cfg)
    if fused and not m.fused_eval:
        pytest.skip("fused engine not enabled yet")
    x = torch.from_numpy(g[f"{tag}_x"]).cuda()
    with torch.no_grad():
        torch.manual_seed(int(g["seed_fwd"]))
        desc, fp, cidx = m(x, use_engine=fused)
    for i in range(3):
        assert np.array_equal(cidx[i].cpu().numpy(), g[f"{tag}_center_idx{i}"])
        assert fp[i].shape[1] == 256 and fp[i].shape[3] == 1
        assert np.allclose(samples(fp[i].contiguous()), g[f"{tag}_fp{i}_samples"], atol=2e-4, rtol=1e-4)
        assert np.allclose(summarize(fp[i].contiguous()), g[f"{tag}_fp{i}_summary"], rtol=2e-5, atol=2e-6)
    _check_desc(desc, g[f"{tag}_desc"])


@pytest.mark.parametrize("tag", ["small", "full"])
@pytest.mark.parametrize("fused", [False, True])
def test_patch_aug_net_backbone_indices_and_sa_features_vs_reference_vectors(tag, fused):
    """The backbone outputs the descriptor test above does not see, against the vectors of the reference's own ``backbone()``
    (patch_aug_net.py:155-192): ``sample_idx_origin`` (:169-177) at every level and the last set-abstraction level's features.
    Module path: bit-equal including the reference's ``randperm`` column order (same CPU generator, same seed).  Engine: it keeps each
    group's nsample nearest in ascending order and never draws the permutation (max-pool is order-invariant), so rows are compared
    as sorted lists -- duplicates included."""
    from patchaugnet_amd import backbone as bb
    from patchaugnet_amd.engine import engine_for
    g = golden("patch_aug_net")
    cfg = configs.patch_aug_net_config()
    if tag == "small":
        cfg = configs.scaled_config(cfg, 512)
    m = _pan(cfg)
    x = torch.from_numpy(g[f"{tag}_x"]).cuda()
    with torch.no_grad():
        if fused:
            if not m.fused_eval:
                pytest.skip("fused engine not enabled yet")
            eng = engine_for(m, x.device)
            eng.keep_geometry = True
            try:
                eng.forward(x)
            finally:
                eng.keep_geometry = False
            geo, eng.last_geometry = eng.last_geometry, None
            c_o, s_o = bb.origin_indices(geo["center_idx"], geo["sample_idx"])
            sa2 = geo["sa_features"][2].transpose(1, 2).contiguous()                 # (B, m2, C) -> the reference's (B, C, m2)
        else:
            torch.manual_seed(int(g["seed_fwd"]))
            res = m.backbone(x.squeeze(1))
            c_o, s_o, sa2 = res["center_idx_origin"], res["sample_idx_origin"], res["sa_features"][2]
    for i in range(3):
        ref = g[f"{tag}_sample_idx{i}"]
        got = s_o[i].cpu().numpy()
        assert got.shape == ref.shape and got.dtype == ref.dtype
        assert np.array_equal(c_o[i].cpu().numpy(), g[f"{tag}_center_idx{i}"])
        if fused:
            assert np.array_equal(np.sort(got, axis=-1), np.sort(ref, axis=-1)), f"level {i}"
        else:
            assert np.array_equal(got, ref), f"level {i}"
    assert np.allclose(samples(sa2), g[f"{tag}_sa2_samples"], atol=2e-4, rtol=1e-4)
    assert np.allclose(summarize(sa2), g[f"{tag}_sa2_summary"], rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize("model_name", ["patch_aug_net", "pptnet"])
def test_opt_in_split_fp16_mode_meets_the_fp32_bar_on_the_reference_vectors(model_name):
    """model.mlp_dtype = "f32x3" (opt-in: the finest FP level's two 256 -> 256 layers from (hi, lo) fp16 operand pairs, csrc/fpx_f32x3.hip) is held
    to the FP32 bar, not the fp16 one: max|d| <= 1e-4 and cosine >= 0.99999 against the vectors of the reference's own classes, indices and
    feature samples as for the fp32 path; its error is reported next to the exact-fp32 engine's on the same input."""
    from patchaugnet_amd import pptnet
    g = golden(model_name)
    if model_name == "patch_aug_net":
        m, ref = _pan(configs.patch_aug_net_config()), g["full_desc"]
    else:
        m = pptnet.Network(param=configs.pptnet_config(), use_normalize=True)
        m.load_state_dict(seeded_sd_from_table("pptnet"), strict=True)
        m, ref = m.cuda().eval(), g["full_desc_l2"]
    x = torch.from_numpy(g["full_x"]).cuda()
    with torch.no_grad():
        d1, _, _ = m(x)
        m.mlp_dtype = "f32x3"
        d3, fp, cidx = m(x)
        assert m._engine.mlp_dtype == "f32x3" and m._engine.fp[0]._premul["x3"] is not None
    for i in range(len(cidx)):
        assert np.array_equal(cidx[i].cpu().numpy(), g[f"full_center_idx{i}"])
    e1, e3 = np.abs(d1.cpu().numpy() - ref).max(), np.abs(d3.cpu().numpy() - ref).max()
    print(f"{model_name}: max|d - reference| exact fp32 {e1:.2e}, split fp16 operands {e3:.2e}; between the two {(d1 - d3).abs().max().item():.2e}")
    _check_desc(d3, ref)
    assert e3 <= 2 * e1 + 2e-6


def test_patch_aug_net_training_tuple_and_backward():
    """forward(x, nn_dict) -> ((desc, patch_recon_data), fp_features, center_idx) (patch_aug_net.py:68-107); gradients flow."""
    g = golden("patch_aug_net")
    cfg = configs.scaled_config(configs.patch_aug_net_config(), 512)
    m = _pan(cfg)
    x = torch.from_numpy(g["small_x"]).cuda()
    torch.manual_seed(int(g["seed_fwd"]))
    (desc, data), fp, cidx = m(x, nn_dict={(0, 1): None})
    assert data["cloud_indices"] == [0, 1] and len(data["reconstructed_patches"]) == 2
    assert np.allclose(samples(data["reconstructed_patches"][0]), g["small_recon0_samples"], atol=2e-4)
    assert np.allclose(samples(data["origin_patches"][0].contiguous()), g["small_origin_patches0_samples"], atol=0)
    _check_desc(desc, g["small_desc"])
    m.train()
    torch.manual_seed(1)
    (desc, data), _, _ = m(x, nn_dict={(0, 1): None})
    (desc.sum() + data["reconstructed_patches"][0].sum()).backward()
    gw = m.backbone.SA_modules[0].mlps[0].layer0.conv.weight.grad
    assert gw is not None and torch.isfinite(gw).all() and gw.abs().sum() > 0


@pytest.mark.parametrize("fused", [False, True])
def test_patch_aug_net_vs_oracle_fresh_inputs(fused):
    from oracle import models_cpu
    from patchaugnet_amd.weights import synthetic_submaps
    cfg = configs.scaled_config(configs.patch_aug_net_config(), 1024)
    m = _pan(cfg)
    if fused and not m.fused_eval:
        pytest.skip("fused engine not enabled yet")
    sd = seeded_sd_from_table("patch_aug_net")
    x = torch.cat([synthetic_submaps(2, 1024, 77, "uniform"), synthetic_submaps(2, 1024, 78, "street")])
    with torch.no_grad():
        torch.manual_seed(5)
        d_ref, fp_ref, c_ref = models_cpu.patch_aug_net_forward(sd, cfg, x)
        torch.manual_seed(5)
        d, fp, c = m(x.cuda(), use_engine=fused)
    for a, b in zip(c, c_ref):
        assert torch.equal(a.cpu(), b)
    _check_desc(d, d_ref.numpy())


@pytest.mark.parametrize("tag", ["small", "full"])
@pytest.mark.parametrize("fused", [False, True])
def test_pptnet_vs_reference_vectors(tag, fused):
    from patchaugnet_amd import pptnet
    g = golden("pptnet")
    cfg = configs.pptnet_config()
    if tag == "small":
        cfg = configs.scaled_config(cfg, 1024)
    sd = seeded_sd_from_table("pptnet")
    x = torch.from_numpy(g[f"{tag}_x"]).cuda()
    for norm, key in ((False, f"{tag}_desc"), (True, f"{tag}_desc_l2")):
        m = pptnet.Network(param=cfg, use_normalize=norm)
        m.load_state_dict(sd, strict=True)
        m = m.cuda().eval()
        with torch.no_grad():
            d, fp, cidx = m(x, use_engine=fused)
        ref = g[key]
        scale = np.abs(ref).max()
        assert np.abs(d.cpu().numpy() - ref).max() <= 2e-4 * scale
        for i in range(4):
            assert np.array_equal(cidx[i].cpu().numpy(), g[f"{tag}_center_idx{i}"])


def test_training_step_quadruplet_plus_patch_chamfer():
    """BASELINE.json configs[3] in miniature: one quadruplet tuple (1 query + 2 positives + 14 negatives + 1 other = 18 clouds,
    configs/patch_aug_net.yaml:60-62) of 512 points, nn_dict with the two (query, positive) pairs -> 3 related clouds;
    quadruplet loss + HIP patch-Chamfer loss, backward through the HIP backward kernels, one Adam step."""
    from patchaugnet_amd import train
    from patchaugnet_amd.weights import synthetic_submaps
    cfg = configs.scaled_config(configs.patch_aug_net_config(), 512)
    m = _pan(cfg)
    x = synthetic_submaps(18, 512, 5).squeeze(1)
    q, pos, neg, oth = x[None, :1], x[None, 1:3], x[None, 3:17], x[None, 17:]
    opt = torch.optim.Adam(m.parameters(), 1e-4)
    before = {k: v.detach().clone() for k, v in m.named_parameters()}
    torch.manual_seed(3)
    out = train.training_step(m, opt, q, pos, neg, oth, nn_dict={(0, 1): None, (0, 2): None}, num_points=512)
    assert set(out) == {"place_recognition", "patch_recon_a2a", "total"} and all(np.isfinite(v) for v in out.values())
    assert out["patch_recon_a2a"] > 0 and abs(out["total"] - out["place_recognition"] - out["patch_recon_a2a"]) < 1e-5
    moved = [k for k, v in m.named_parameters() if not torch.equal(v.detach(), before[k])]
    for part in ("backbone.SA_modules.0", "backbone.FP_modules.0", "aggregation.vlads.2", "aggregation.afa.fc", "decoder.fc3"):
        assert any(k.startswith(part) for k in moved), part


@pytest.mark.parametrize("model_name", ["patch_aug_net", "pptnet"])
def test_fused_engine_is_batch_size_invariant(model_name):
    """evaluate.py:170 extracts with batch 100; BASELINE's bench uses 32.  Every submap is independent in eval mode, and the
    engine's per-row / per-cloud arithmetic does not depend on how clouds are batched: descriptors must be bit-identical for
    B = 1, 3, 32, 100 (heads with B > 64 take the chunked split-K path)."""
    from patchaugnet_amd import pptnet
    from patchaugnet_amd.weights import synthetic_submaps
    if model_name == "patch_aug_net":
        m = _pan(configs.patch_aug_net_config())
    else:
        m = pptnet.Network(param=configs.pptnet_config(), use_normalize=True)
        m.load_state_dict(seeded_sd_from_table("pptnet"), strict=True)
        m = m.cuda().eval()
    x = torch.cat([synthetic_submaps(60, 4096, 11, "uniform"), synthetic_submaps(40, 4096, 12, "street")]).cuda()
    with torch.no_grad():
        full = m(x, return_feat=False)
        assert full.shape == (100, 256) and torch.isfinite(full).all()
        for lo, hi in ((0, 1), (5, 8), (20, 52), (60, 100)):
            part = m(x[lo:hi].contiguous(), return_feat=False)
            assert torch.equal(part, full[lo:hi]), (lo, hi, (part - full[lo:hi]).abs().max().item())


@pytest.mark.parametrize("npts,sampling", [(2048, [512, 64, 16]), (8192, [2048, 256, 32]), (3000, [700, 100, 20])])
def test_fused_engine_matches_module_path_at_other_cloud_sizes(npts, sampling):
    """The engine is not specialised to 4096 points: other sizes (the pruned kNN's lower bound 2048, clouds above its 4096 limit,
    a size that is no multiple of anything) give the module path's descriptors and bit-identical centre indices."""
    from patchaugnet_amd.weights import synthetic_submaps
    cfg = configs.patch_aug_net_config()
    cfg["NUM_POINTS"], cfg["SAMPLING"], cfg["MAX_SAMPLES"] = npts, sampling, [sampling[1], sampling[0], npts]
    m = _pan(cfg)
    x = torch.cat([synthetic_submaps(2, npts, 41, "uniform"), synthetic_submaps(1, npts, 42, "street")]).cuda()
    with torch.no_grad():
        torch.manual_seed(3)
        d_mod, fp_mod, c_mod = m(x, use_engine=False)
        d_eng, fp_eng, c_eng = m(x, use_engine=True)
    for a, b in zip(c_eng, c_mod):
        assert torch.equal(a, b)
    _check_desc(d_eng, d_mod.cpu().numpy())
    for a, b in zip(fp_eng, fp_mod):
        assert (a - b).abs().max().item() <= 2e-4 * max(b.abs().max().item(), 1.0)


def test_forward_with_precomputed_geometry_equals_the_plain_forward():
    """backbone.geometry(x) (sampling, neighbour search with the groupers' permutation, 3-NN weights) handed to forward(geometry=...) gives
    exactly the forward that computes them itself: module path, train mode statistics, patch-reconstruction branch included."""
    from patchaugnet_amd import configs, patch_aug_net, pointops
    from patchaugnet_amd.weights import seeded_state_dict, synthetic_submaps
    n = 1024
    cfg = configs.scaled_config(configs.patch_aug_net_config(), n)
    m = patch_aug_net.Network(param=cfg, use_a2a_recon=True, use_l2_norm=True)
    m.load_state_dict(seeded_state_dict(m.state_dict()))
    m = m.cuda().train()
    x = synthetic_submaps(4, n, seed=3).cuda()
    groupers = [g for g in m.modules() if isinstance(g, pointops.QueryAndGroup_Edge) and g.radius is None and g.knn_dilation > 1]
    for g in groupers:
        g.perm_buffer = torch.randperm(g.nsample).cuda()          # the same permutation in both forwards
    nn_dict = {(0, 1): None, (0, 2): None}
    try:
        sd = {k: v.clone() for k, v in m.state_dict().items()}
        with torch.no_grad():
            (d0, r0), fp0, c0 = m(x, nn_dict)
            m.load_state_dict(sd)                                 # undo the running-statistics update of the first forward
            geo = m.backbone.geometry(x.squeeze(1))
            (d1, r1), fp1, c1 = m(x, nn_dict, geometry=geo)
    finally:
        for g in groupers:
            g.perm_buffer = None
    # indices and pure gathers: exactly equal; dense outputs: train-mode BatchNorm statistics are summed with atomics, so two forwards of the
    # SAME kind already differ in the last bits
    assert all(torch.equal(a, b) for a, b in zip(c0, c1))
    assert all(torch.equal(a, b) for a, b in zip(r0["origin_patches"], r1["origin_patches"]))
    close = lambda a, b: (a - b).abs().max().item() <= 1e-5 * max(1.0, a.abs().max().item())
    assert close(d0, d1) and all(close(a, b) for a, b in zip(fp0, fp1))
    assert all(close(a, b) for a, b in zip(r0["reconstructed_patches"], r1["reconstructed_patches"]))
