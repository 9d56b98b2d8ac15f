"""The CPU model restatement (oracle/models_cpu.py) against vectors produced by the
reference's own Python classes (tests/golden/, see oracle/gen_golden.py).  No GPU."""
import numpy as np
import pytest
import torch

from oracle import models_cpu
from patchaugnet_amd import configs
from tests._util import golden, seeded_sd_from_table, summarize, samples


@pytest.mark.parametrize("tag", ["small", "full"])
def test_patch_aug_net_matches_reference_vectors(tag):
    g = golden("patch_aug_net")
    cfg = configs.patch_aug_net_config()
    if tag == "small":
        cfg = configs.scaled_config(cfg, 512)
    sd = seeded_sd_from_table("patch_aug_net")
    x = torch.from_numpy(g[f"{tag}_x"])
    with torch.no_grad():
        torch.manual_seed(int(g["seed_fwd"]))
        desc, fp, cidx = models_cpu.patch_aug_net_forward(sd, cfg, x)
        torch.manual_seed(int(g["seed_fwd"]))
        res = models_cpu.patch_aug_net_backbone(sd, cfg, x.squeeze(1))
    assert np.abs(desc.numpy() - g[f"{tag}_desc"]).max() <= 1e-6
    for i in range(3):
        assert np.array_equal(cidx[i].numpy(), g[f"{tag}_center_idx{i}"])
        assert np.array_equal(res["sample_idx_origin"][i].numpy(), g[f"{tag}_sample_idx{i}"])
        assert np.allclose(samples(fp[i]), g[f"{tag}_fp{i}_samples"], atol=1e-5)
        assert np.allclose(summarize(fp[i]), g[f"{tag}_fp{i}_summary"], rtol=1e-6, atol=1e-7)
    assert np.allclose(samples(res["sa_features"][2]), g[f"{tag}_sa2_samples"], atol=1e-5)
    # decoder on cloud 0's 1024 patch features (patch_aug_net.py:83-98)
    with torch.no_grad():
        feats = torch.nn.functional.normalize(fp[1][0].squeeze(-1).transpose(1, 0))
        rec = models_cpu.patch_decoder(sd, feats)
    assert np.allclose(samples(rec), g[f"{tag}_recon0_samples"], atol=1e-5)


@pytest.mark.parametrize("tag", ["small", "full"])
def test_pptnet_matches_reference_vectors(tag):
    g = golden("pptnet")
    cfg = configs.pptnet_config()
    if tag == "small":
        cfg = configs.scaled_config(cfg, 1024)
    sd = seeded_sd_from_table("pptnet")
    x = torch.from_numpy(g[f"{tag}_x"])
    with torch.no_grad():
        d, fp, cidx = models_cpu.pptnet_forward(sd, cfg, x, use_normalize=False)
        d2 = models_cpu.pptnet_forward(sd, cfg, x, use_normalize=True, return_feat=False)
    ref = g[f"{tag}_desc"]
    assert np.abs(d.numpy() - ref).max() <= 1e-6 * np.abs(ref).max()
    assert np.abs(d2.numpy() - g[f"{tag}_desc_l2"]).max() <= 1e-6
    for i in range(4):
        assert np.array_equal(cidx[i].numpy(), g[f"{tag}_center_idx{i}"])
        assert np.allclose(samples(fp[i]), g[f"{tag}_fp{i}_samples"], atol=1e-5)


@pytest.mark.parametrize("agg_type,gating", [(0, False), (0, True), (2, False), (2, True), (3, False), (3, True)])
def test_aggregation_heads_match_reference_vectors(agg_type, gating):
    """oracle.models_cpu.netvlad_base + spvlad_aggregate against the reference's SpatialPyramidNetVLAD for every aggregation type the
    product's HIP heads cover (tests/golden/heads.npz, oracle/gen_head_golden.py)."""
    from oracle.gen_head_golden import KS, NS, features
    from patchaugnet_amd import loupe
    from patchaugnet_amd.weights import seeded_state_dict
    g = golden("heads")
    agg = loupe.SpatialPyramidNetVLAD([256] * 3, NS, KS, [256] * 3, gating=gating, aggregation_type=agg_type)
    sd = seeded_state_dict(agg.state_dict(), seed=100 + agg_type)
    assert sorted(sd.keys()) == g[f"type{agg_type}_gating{int(gating)}_keys"].tolist()        # same parameter names as the reference class
    feats = features()
    sdp = {"a." + k: v for k, v in sd.items()}
    with torch.no_grad():
        v = torch.cat([models_cpu.netvlad_base(sdp, f"a.vlads.{i}", feats[i], NS[i], 256, KS[i]) for i in range(3)], dim=-1)
        d = models_cpu.spvlad_aggregate(sdp, "a", v, agg_type, gating)
        agg.load_state_dict(sd)
        d_mod = agg.eval()(feats)                                                              # the product's module path too
    ref = g[f"type{agg_type}_gating{int(gating)}"]
    assert np.abs(d.numpy() - ref).max() <= 1e-6 and np.abs(d_mod.numpy() - ref).max() <= 1e-6
