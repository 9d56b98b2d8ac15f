"""pa_netvlad / pa_afa / the FC, max-pool and gated heads against the CPU oracle (oracle/models_cpu.py: netvlad_base,
adaptive_feature_aggregator, spvlad_aggregate) and the torch module path (patchaugnet_amd/loupe.py).  fp32 tolerance."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _seed_module(m, seed):
    from patchaugnet_amd.weights import seeded_state_dict
    m.load_state_dict(seeded_state_dict(m.state_dict(), seed=seed))
    return m.cuda().eval()


@pytest.mark.parametrize("b,n,k", [(2, 4096, 64), (3, 1024, 16), (2, 128, 4), (1, 100, 7), (2, 700, 33), (1, 2048, 64)])
def test_netvlad_scale(b, n, k):
    from patchaugnet_amd import loupe
    from patchaugnet_amd.engine import _Vlad
    v = _seed_module(loupe.NetVLADBase(256, n, k, 256, gating=False), seed=n + k)
    x = torch.randn(b, n, 256, device="cuda") * 0.7
    with torch.no_grad():
        ref = v(x.transpose(1, 2).unsqueeze(-1))                         # (B, 256, K)
        out = torch.full((b, 256, k + 5), 7.0, device="cuda")
        _Vlad(v, x.device).run(x, out, k + 5, 3)
        out_t = torch.full((b, k + 5, 256), 7.0, device="cuda")
        _Vlad(v, x.device).run(x, out_t, k + 5, 3, rows=True)                              # cluster-major rows (B, sum K, C)
    got = out[:, :, 3:3 + k]
    assert torch.all(out[:, :, :3] == 7.0) and torch.all(out[:, :, 3 + k:] == 7.0)       # neighbours of the block untouched
    err = (got - ref).abs().max().item()
    assert err <= 2e-5, err
    from oracle import models_cpu
    sd = {"v." + kk: t.cpu() for kk, t in v.state_dict().items()}
    orc = models_cpu.netvlad_base(sd, "v", x.cpu().transpose(1, 2).unsqueeze(-1), n, 256, k)          # loupe.py:191-222 restated on the CPU
    assert (got.cpu() - orc).abs().max().item() <= 2e-5
    assert torch.equal(out_t[:, 3:3 + k].transpose(1, 2), got) and torch.all(out_t[:, :3] == 7.0) and torch.all(out_t[:, 3 + k:] == 7.0)


@pytest.mark.parametrize("b,ktot", [(32, 84), (2, 84), (5, 21), (17, 100), (70, 84)])
def test_afa(b, ktot):
    from patchaugnet_amd import loupe
    from patchaugnet_amd.engine import _Afa
    afa = _seed_module(loupe.AdaptiveFeatureAggregator(256, ktot, 256), seed=ktot)
    v = torch.nn.functional.normalize(torch.randn(b, 256, ktot, device="cuda"), dim=1)
    with torch.no_grad():
        ref = afa(v).squeeze(-1)
        eng = _Afa(afa, v.device)
        got = eng.run(v.contiguous())
        got_rows = eng.run_rows(v.transpose(1, 2).contiguous())                            # cluster-major, five launches
        got_fused = eng.run_fused(v.transpose(1, 2).contiguous())                          # cluster-major, two launches: what the engine runs
    err = (got - ref).abs().max().item()
    assert err <= 2e-5, err
    assert (got_rows - ref).abs().max().item() <= 2e-5
    assert (got_fused - ref).abs().max().item() <= 2e-5 and (got_fused - got_rows).abs().max().item() <= 1e-5
    import copy
    ref64 = copy.deepcopy(afa).cpu().double()(v.cpu().double()).squeeze(-1)               # the fused form is as close to fp64 as the others
    assert (got_fused.cpu().double() - ref64).abs().max().item() <= 1.5 * max((got_rows.cpu().double() - ref64).abs().max().item(), 2e-7)
    from oracle import models_cpu
    orc = models_cpu.adaptive_feature_aggregator({"a." + kk: t.cpu() for kk, t in afa.state_dict().items()}, "a", v.cpu())
    assert (got.cpu() - orc).abs().max().item() <= 2e-5 and (got_rows.cpu() - orc).abs().max().item() <= 2e-5
    assert (got_fused.cpu() - orc).abs().max().item() <= 2e-5


def test_afa_fused_attention_extremes():
    """The soft-max over clusters inside the fused head: one cluster with a dominant logit (w -> 1 for it, ~0 elsewhere), all-negative
    rows (relu kills them) and a zero row -- against the module in fp64."""
    from patchaugnet_amd import loupe
    from patchaugnet_amd.engine import _Afa
    afa = _seed_module(loupe.AdaptiveFeatureAggregator(256, 84, 256), seed=5)
    v = torch.nn.functional.normalize(torch.randn(6, 256, 84, device="cuda"), dim=1)
    v[0, :, 7] *= 40.0
    v[1, :, :] = -v[1].abs()
    v[2, :, 3] = 0.0
    v[3] = 0.0
    import copy
    with torch.no_grad():
        got = _Afa(afa, v.device).run_fused(v.transpose(1, 2).contiguous())
        ref = copy.deepcopy(afa).cpu().double()(v.cpu().double()).squeeze(-1)
    assert torch.isfinite(got).all()
    assert (got.cpu().double() - ref).abs().max().item() <= 2e-5


@pytest.mark.parametrize("b,scales", [(3, [(128, 4), (1024, 16), (4096, 64)]), (2, [(64, 1), (256, 4), (1024, 16), (4096, 64)]),
                                      (2, [(100, 7), (700, 33)]), (1, [(2048, 64)]), (33, [(128, 4), (1024, 16)])])
def test_netvlad_pyramid_is_bit_identical_to_the_per_scale_calls(b, scales):
    """pa_netvlad_pyramid (coarse scales in one accumulate launch, every scale in one finalize launch) = pa_netvlad_rows per scale, bit for bit."""
    from patchaugnet_amd import loupe
    from patchaugnet_amd.engine import _Pyramid, _Vlad
    vl, xs = [], []
    for n, k in scales:
        v = _seed_module(loupe.NetVLADBase(256, n, k, 256, gating=False), seed=n + k)
        vl.append(_Vlad(v, torch.device("cuda")))
        xs.append(torch.randn(b, n, 256, device="cuda") * 0.7)
    ktot = sum(k for _, k in scales)
    ref = torch.full((b, ktot, 256), 7.0, device="cuda")
    koff = 0
    for v, x in zip(vl, xs):
        v.run(x, ref, ktot, koff, rows=True)
        koff += v.k
    got = torch.full((b, ktot, 256), -3.0, device="cuda")
    _Pyramid(vl).run(xs, got)
    assert torch.equal(got, ref)


@pytest.mark.parametrize("agg_type,gating", [(0, False), (0, True), (2, True), (3, False), (3, True)])
def test_other_aggregation_heads_run_on_hip_kernels(agg_type, gating):
    """SpatialPyramidNetVLAD aggregation types 0 (FC) and 3 (max-pool) and context gating (loupe.py:289-326) go through the HIP head
    kernels of the fused engine (no torch fallback): whole-model descriptors against the CPU oracle and the module path."""
    from oracle import models_cpu
    from patchaugnet_amd import configs, patch_aug_net
    from patchaugnet_amd.weights import seeded_state_dict, synthetic_submaps
    cfg = configs.scaled_config(configs.patch_aug_net_config(), 512)
    cfg["AGGREGATION_TYPE"], cfg["GATING"] = agg_type, gating
    m = patch_aug_net.Network(param=cfg, use_a2a_recon=False, use_l2_norm=True)
    sd = seeded_state_dict(m.state_dict(), seed=40 + agg_type)
    m.load_state_dict(sd)
    m = m.cuda().eval()
    x = synthetic_submaps(5, 512, seed=3)
    with torch.no_grad():
        torch.manual_seed(0)
        d_eng = m(x.cuda(), return_feat=False)
        kind = m._engine.head_kind
        torch.manual_seed(0)
        d_mod = m(x.cuda(), return_feat=False, use_engine=False)
        torch.manual_seed(0)
        d_orc = models_cpu.patch_aug_net_forward(sd, cfg, x, return_feat=False)
    assert kind == {0: "fc", 2: "afa", 3: "max"}[agg_type] and (m._engine.gate is not None) == gating
    assert (d_eng.cpu() - d_orc).abs().max().item() <= 1e-4
    assert (d_eng - d_mod).abs().max().item() <= 1e-4


def test_unsupported_head_shape_raises_instead_of_falling_back():
    from patchaugnet_amd import configs, patch_aug_net
    from patchaugnet_amd.weights import synthetic_submaps
    cfg = configs.scaled_config(configs.patch_aug_net_config(), 512)
    cfg["CLUSTER_SIZE"] = [4, 16, 80]                           # more clusters per scale than the NetVLAD kernel is built for
    m = patch_aug_net.Network(param=cfg, use_a2a_recon=False, use_l2_norm=True).cuda().eval()
    with torch.no_grad(), pytest.raises(ValueError, match="fused engine"):
        m(synthetic_submaps(2, 512, seed=1).cuda(), return_feat=False)
