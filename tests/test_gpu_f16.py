"""fp16-operand chain kernels (mlp_chain_f16.hip; BASELINE.json configs[4] "fp16 MFMA MLP path"): fp32 accumulation and outputs,
fp16 weights / activations inside the kernel.  Tolerance per SURVEY.md section 8: cosine >= 0.999 on descriptors; indices exact."""
import numpy as np
import pytest
import torch

from patchaugnet_amd import configs
from tests._util import golden, seeded_sd_from_table

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("rows,k,n,relu,res", [(1000, 64, 128, 0, False), (77, 259, 256, 1, True), (4096, 512, 1024, 0, False), (33, 20, 16, 1, False),
                                               (512, 512, 1024, 0, False), (512, 512, 512, 1, True), (2048, 256, 512, 0, True), (100, 128, 256, 1, True),
                                               (131072, 256, 256, 1, False)])
def test_pa_linear_f16(rows, k, n, relu, res):
    from patchaugnet_amd._lib import call, ptr
    from patchaugnet_amd.engine import pack_weights_f16
    x = torch.randn(rows, k, device="cuda")
    w = torch.randn(n, k, device="cuda") / k ** 0.5
    bias = torch.randn(n, device="cuda")
    r = torch.randn(rows, n, device="cuda") if res else None
    kpad = (k + 3) // 4 * 4
    wt = torch.zeros(kpad, n, device="cuda")
    wt[:k] = w.t()
    out = torch.empty(rows, n, device="cuda")
    call("pa_linear_f16", rows, k, n, ptr(x), k, ptr(wt), ptr(pack_weights_f16(wt)), ptr(bias), relu, ptr(r), n if res else 0, ptr(out), n)
    # reference with the same operand rounding: fp16(x) . fp16(w) accumulated exactly, fp32 bias
    ref = x.half().double() @ w.half().double().t() + bias.double()
    if relu:
        ref = ref.clamp_min(0)
    if res:
        ref = ref + r.double()
    err = (out.double() - ref).abs().max().item()
    assert err <= 1e-4 * max(ref.abs().max().item(), 1.0), err          # only the fp32 accumulation order differs
    full = x.double() @ w.double().t() + bias.double()                   # and against unrounded operands: fp16 rounding level
    if relu:
        full = full.clamp_min(0)
    if res:
        full = full + r.double()
    assert (out.double() - full).abs().max().item() <= 4e-3 * max(full.abs().max().item(), 1.0)


def _cos(a, b):
    return (a * b).sum(1) / (np.linalg.norm(a, axis=1) * np.linalg.norm(b, axis=1))


@pytest.mark.parametrize("tag", ["small", "full"])
def test_patch_aug_net_f16_mlp_path(tag):
    from patchaugnet_amd import patch_aug_net
    g = golden("patch_aug_net")
    cfg = configs.patch_aug_net_config()
    if tag == "small":
        cfg = configs.scaled_config(cfg, 512)
    m = patch_aug_net.Network(param=cfg, use_a2a_recon=True, use_l2_norm=True)
    m.load_state_dict(seeded_sd_from_table("patch_aug_net"), strict=True)
    m = m.cuda().eval()
    m.mlp_dtype = "f16"
    x = torch.from_numpy(g[f"{tag}_x"]).cuda()
    with torch.no_grad():
        desc, fp, cidx = m(x)
        assert m._engine.mlp_dtype == "f16"
        m.mlp_dtype = "f32"
        d32, _, _ = m(x)                                       # the engine is rebuilt for the other dtype
        assert m._engine.mlp_dtype == "f32"
    for i in range(3):
        assert np.array_equal(cidx[i].cpu().numpy(), g[f"{tag}_center_idx{i}"])      # sampling is untouched by the MLP dtype
    d = desc.cpu().numpy()
    assert np.isfinite(d).all()
    assert _cos(d, g[f"{tag}_desc"]).min() >= 0.999
    assert np.abs(d - g[f"{tag}_desc"]).max() <= 5e-3
    assert np.abs(d32.cpu().numpy() - g[f"{tag}_desc"]).max() <= 1e-4


@pytest.mark.parametrize("tag", ["small", "full"])
def test_pptnet_f16_mlp_path(tag):
    """BASELINE.json configs[4]: PPT-Net with the fp16 MFMA MLP path."""
    from patchaugnet_amd import pptnet
    g = golden("pptnet")
    cfg = configs.pptnet_config()
    if tag == "small":
        cfg = configs.scaled_config(cfg, 1024)
    m = pptnet.Network(param=cfg, use_normalize=True)
    m.load_state_dict(seeded_sd_from_table("pptnet"), strict=True)
    m = m.cuda().eval()
    m.mlp_dtype = "f16"
    with torch.no_grad():
        d, fp, cidx = m(torch.from_numpy(g[f"{tag}_x"]).cuda())
    for i in range(4):
        assert np.array_equal(cidx[i].cpu().numpy(), g[f"{tag}_center_idx{i}"])
    assert _cos(d.cpu().numpy(), g[f"{tag}_desc_l2"]).min() >= 0.999


@pytest.mark.parametrize("split", [1, 0])
@pytest.mark.parametrize("b,n,c", [(2, 1024, 64), (3, 256, 128), (2, 64, 256), (1, 20, 64), (2, 100, 128), (1, 333, 64), (2, 37, 256)])
def test_sa_attention_f16_against_the_fp32_kernel(b, n, c, split):
    """pa_sa_attention_f16 (csrc/attention_f16.hip: both contractions on fp16 MFMA, fp32 soft-max) against pa_sa_attention on the same
    [Y | V] rows: ragged n (not a multiple of 16 / 32 / the tile), every width it is built for.  split = 1 carries the energy operands as
    (hi, lo) fp16 pairs: only V and the soft-max values are rounded to fp16 (2^-11 each); split = 0 also rounds the logits' operands."""
    from patchaugnet_amd import _lib
    from patchaugnet_amd._lib import call, ptr
    g = torch.Generator().manual_seed(n * c + split)
    yv = (torch.randn(b * n, 2 * c, generator=g) * 0.7).cuda()
    x = torch.randn(b * n, c, generator=g).cuda()
    stats = torch.empty(b * n, 2, device="cuda")
    d32 = torch.empty(b * n, c, device="cuda")
    call("pa_sa_attention", b, n, c, ptr(yv), ptr(x), ptr(stats), ptr(d32))
    d16 = torch.full((b * n, c), float("nan"), device="cuda")
    scratch = torch.empty(_lib.lib().pa_sa_attention_f16_scratch_halfs(b, n, c, split), dtype=torch.float16, device="cuda")
    call("pa_sa_attention_f16", b, n, c, split, ptr(yv), ptr(x), ptr(scratch), ptr(stats), ptr(d16))
    xr32, xr16 = (x - d32).double(), (x - d16).double()
    assert torch.isfinite(d16).all()
    scale = xr32.abs().max().item()
    err = (xr16 - xr32).abs().max().item()
    assert err <= (2e-3 if split else 3e-2) * scale, (err, scale)


@pytest.mark.parametrize("b,n,c", [(2, 1024, 64), (3, 256, 128), (2, 64, 256), (1, 20, 64), (2, 37, 256)])
def test_sa_attention_f16_fused_trans_layer(b, n, c):
    """pa_sa_attention_trans_f16: the layer behind the attention (trans_conv + folded BatchNorm + ReLU + residual) in the second pass's epilogue
    against the two-launch form (pa_sa_attention_f16, then pa_linear_f16 on d with the residual): same fp16 operands, another summation order."""
    from patchaugnet_amd import _lib
    from patchaugnet_amd._lib import call, ptr
    from patchaugnet_amd.engine import pack_weights_f16
    g = torch.Generator().manual_seed(n + c)
    yv = (torch.randn(b * n, 2 * c, generator=g) * 0.7).cuda()
    x = torch.randn(b * n, c, generator=g).cuda()
    wt = (torch.randn(c, c, generator=g) / c ** 0.5).cuda()          # K-major
    bt = torch.randn(c, generator=g).cuda()
    stats = torch.empty(b * n, 2, device="cuda")
    d = torch.empty(b * n, c, device="cuda")
    scratch = torch.empty(_lib.lib().pa_sa_attention_f16_scratch_halfs(b, n, c, 1), dtype=torch.float16, device="cuda")
    call("pa_sa_attention_f16", b, n, c, 1, ptr(yv), ptr(x), ptr(scratch), ptr(stats), ptr(d))
    two = torch.empty(b * n, c, device="cuda")
    call("pa_linear_f16", b * n, c, c, ptr(d), c, ptr(wt), ptr(pack_weights_f16(wt)), ptr(bt), 1, ptr(x), c, ptr(two), c)
    wtp = torch.empty(c * c, dtype=torch.float16, device="cuda")
    call("pa_sa_attention_f16_pack_trans", c, ptr(wt), ptr(wtp))
    one = torch.full((b * n, c), float("nan"), device="cuda")
    call("pa_sa_attention_trans_f16", b, n, c, 1, ptr(yv), ptr(x), ptr(scratch), ptr(stats), ptr(wtp), ptr(bt), ptr(one))
    assert torch.isfinite(one).all()
    assert (one - two).abs().max().item() <= 1e-4 * max(1.0, two.abs().max().item())


@pytest.mark.parametrize("mode,g16", [(8, True), (4, True), (8, False), (4, False), (0, False)])
@pytest.mark.parametrize("B,n,m,c1", [(3, 4096, 1024, 3), (1, 1000, 256, 3), (2, 77, 19, 1), (1, 300, 64, 4)])
def test_finest_fp_level_f16_on_lds_shared_weights(mode, g16, B, n, m, c1, monkeypatch):
    """The finest FP level of the fp16 path (skip = xyz, two 256 -> 256 layers left): fpx_f16.hip shares the weights of a workgroup's waves
    through LDS and keeps the activations in registers (mode 8 / 4 = waves per workgroup; g16 = the pre-multiplied features as an fp16 table
    written by pa_fp_premul_g16, the engine's default; otherwise pa_linear_f16 + pa_fp_chain_premul_f16 with an fp32 table); mode 0 = the
    wave-private LDS-tile kernel.  Reference: the same operand roundings in float64 (fp16 weights, fp16 hidden activations, fp16 table when
    g16; everything else exact); a hidden value within rounding distance of an fp16 tie may round the other way, hence 2e-3 of the tensor's
    scale; rows ragged against the 128 / 256-row workgroup tile."""
    from patchaugnet_amd import _lib
    from patchaugnet_amd.engine import _Chain
    from tests.test_gpu_chain import make_layers
    c2 = 256
    ref, eng = make_layers([c2 + c1, 256, 256, 256], seed=11 + c1)
    g = torch.Generator().manual_seed(n)
    known = torch.randn(B, m, c2, generator=g)
    skip = torch.randn(B, n, c1, generator=g)
    idx3 = torch.randint(0, m, (B, n, 3), generator=g).int()
    w3 = torch.rand(B, n, 3, generator=g)
    w3 = (w3 / w3.sum(-1, keepdim=True)).contiguous()
    monkeypatch.setenv("PA_ENGINE_FPX16", "1" if g16 else "0")
    ch = _Chain(eng, f16=True)
    ch.build_premul(c2, c1)
    assert ch._premul["g16"] == g16
    _lib.lib().pa_fpx16_enable(mode)
    try:
        got = ch.fp_premul(known.cuda(), idx3.cuda(), w3.cuda(), skip.cuda(), B, n, m, c2, c1)
        torch.cuda.synchronize()
    finally:
        _lib.lib().pa_fpx16_enable(-1)
    h16 = lambda t: t.half().double()
    (w1, b1), (w2, b2), (w3_, b3) = [(w.float().double(), b.float().double()) for w, b in ref]
    gk = (h16(known.double()) @ h16(w1[:, :c2]).t()).float().double()                    # the pre-multiply: fp16 operands, fp32 accumulation
    if g16:
        gk = h16(gk)
    bi = torch.arange(B)[:, None]
    interp = sum(w3[..., t:t + 1].double() * gk[bi, idx3[:, :, t].long()] for t in range(3))
    h1 = torch.relu(interp + skip.double() @ w1[:, c2:].t() + b1)
    h2 = torch.relu(h16(h1) @ h16(w2).t() + b2)
    exp = torch.relu(h16(h2) @ h16(w3_).t() + b3).reshape(B * n, -1)
    err = (got.double().cpu() - exp).abs().max().item()
    assert err <= 2e-3 * exp.abs().max().item(), (err, exp.abs().max().item())
    full = torch.relu(torch.relu(torch.relu(torch.cat([sum(w3[..., t:t + 1].double() * known.double()[bi, idx3[:, :, t].long()] for t in range(3)),
                                                       skip.double()], -1) @ w1.t() + b1) @ w2.t() + b2) @ w3_.t() + b3).reshape(B * n, -1)
    assert (got.double().cpu() - full).abs().max().item() <= 8e-3 * full.abs().max().item()     # and against unrounded operands


@pytest.mark.parametrize("mag", [0.7, 40.0])
@pytest.mark.parametrize("b,scales", [(3, [(128, 4), (1024, 16), (4096, 64)]), (2, [(64, 1), (256, 4), (1024, 16), (4096, 64)]), (1, [(2048, 64)]),
                                      (2, [(1000, 50)]), (5, [(300, 64)])])
def test_netvlad_pyramid_f16_against_the_fp32_kernel(b, scales, mag):
    """pa_netvlad_pyramid_f16: the 49..64-cluster scale on the 16-bit MFMAs (vlad.hip vlad_accum16_kernel: logits from (hi, lo) fp16 operand pairs,
    aggregation with features and soft-assignments in bf16, fp32 accumulation, everything else fp32) against the fp32 pyramid: every
    intra-normalised cluster row within cosine 0.9999; the other scales run the fp32 kernel in both and are bit-identical.  Ragged point counts
    (1000, 300: partial tiles and a partial last workgroup), a cluster count below 64, and features of magnitude 40: logits of standard deviation
    ~15, nearly one-hot assignments.  There a plain fp16 rounding of the logits' operands flips assignments, and fp16 assignments lose the clusters
    nobody is assigned to -- their masses are 1e-8 and below, under fp16's range, yet intra-normalisation gives their rows the weight of every
    other row (first builds: descriptor cosine 0.52; row cosines near 0 for those clusters)."""
    from patchaugnet_amd import loupe
    from patchaugnet_amd.engine import _Pyramid, _Vlad
    from tests.test_gpu_head import _seed_module
    vl, xs = [], []
    for n, k in scales:
        v = _seed_module(loupe.NetVLADBase(256, n, k, 256, gating=False), seed=n + k)
        vl.append(_Vlad(v, torch.device("cuda")))
        gen = torch.Generator().manual_seed(1000 * b + n + k)
        xs.append((torch.relu(torch.randn(b, n, 256, generator=gen)) * mag).cuda())   # mag = 40: logits of standard deviation ~15, nearly one-hot assignments
    ktot = sum(k for _, k in scales)
    ref = torch.full((b, ktot, 256), 7.0, device="cuda")
    got = torch.full((b, ktot, 256), -3.0, device="cuda")
    _Pyramid(vl).run(xs, ref)
    p16 = _Pyramid(vl, f16=True)
    assert p16.f16
    p16.run(xs, got)
    torch.cuda.synchronize()
    koff = 0
    for n, k in scales:
        r, g = ref[:, koff:koff + k].double(), got[:, koff:koff + k].double()
        if k > 48:
            cos = torch.nn.functional.cosine_similarity(r, g, dim=2)
            whole = torch.nn.functional.cosine_similarity(r.reshape(b, -1), g.reshape(b, -1), dim=1)
            print(f"n={n} k={k} mag={mag}: row cosine min {cos.min().item():.6f} mean {cos.mean().item():.6f}; whole scale min {whole.min().item():.6f}")
            # rows the intra-normalisation left (near) zero -- a cluster whose total mass underflows even fp32: ||v|| < 1e-12 -- have no direction
            live = r.norm(dim=2) > 0.5
            assert live.float().mean().item() > 0.9
            assert cos[live].min().item() >= 0.9999, cos[live].min().item()
            assert (r - g).abs().max().item() <= 2e-2
            assert whole.min().item() >= 0.99999, whole.min().item()
            assert not torch.equal(r, g)
        else:
            assert torch.equal(r, g)
        koff += k


@pytest.mark.parametrize("name", ["patch_aug_net", "pptnet"])
def test_descriptor_only_call_keeps_the_finest_map_in_fp16(name):
    """fp16 path, ``model(x, return_feat=False)`` at full size: the finest feature-propagation level writes its (B x 4096, 256) map as fp16 rows
    (pa_fp_chain_premul_g16h) and the 64-cluster NetVLAD kernel reads those (pa_netvlad_pyramid_f16h) -- half the bytes of the step's largest
    store and of its re-read.  Held to the fp16 path's own bar against the vectors of the reference's classes (cosine >= 0.999), and to the
    return_feat=True form of the same path (fp32 map between the two kernels) at cosine >= 0.9999."""
    from patchaugnet_amd import patch_aug_net, pptnet
    g = golden(name)
    if name == "patch_aug_net":
        m = patch_aug_net.Network(param=configs.patch_aug_net_config(), use_a2a_recon=True, use_l2_norm=True)
        ref = g["full_desc"]
    else:
        m = pptnet.Network(param=configs.pptnet_config(), use_normalize=True)
        ref = g["full_desc_l2"]
    m.load_state_dict(seeded_sd_from_table(name), strict=True)
    m = m.cuda().eval()
    m.mlp_dtype = "f16"
    x = torch.from_numpy(g["full_x"]).cuda()
    with torch.no_grad():
        d_feat, fp, _ = m(x)                                   # fp32 map (the caller gets it)
        eng = m._engine
        seen = []
        orig = eng.pyramid.launch
        eng.pyramid.launch = lambda st, feats, out, phases, x16_mask=0: (seen.append((x16_mask, [None if f is None else f.dtype for f in feats])),
                                                                        orig(st, feats, out, phases, x16_mask=x16_mask))[1]
        try:
            d_only = m(x, return_feat=False)
        finally:
            eng.pyramid.launch = orig
    d_only = d_only[0] if isinstance(d_only, (tuple, list)) else d_only
    assert fp[-1].dtype == torch.float32
    assert any(mask and dts[-1] == torch.float16 for mask, dts in seen), seen      # the fp16 map really was what the NetVLAD kernel read
    a, b = d_only.cpu().numpy(), d_feat.cpu().numpy()
    assert np.isfinite(a).all()
    assert _cos(a, ref).min() >= 0.999, _cos(a, ref).min()
    assert _cos(a, b).min() >= 0.9999, _cos(a, b).min()


@pytest.mark.parametrize("victim", ["sampling", "knn"])
def test_index_kernels_beside_fp16_chain_kernels_on_another_stream(victim):
    """The gfx950 fault of DESIGN.md section 5 (packed fp32 with operand modifiers beside 16x16x32 MFMA kernels), at the op level: the sampling and the kNN
    search must return their serial results while pa_linear_f16 loops on another stream.  Before the fix the sampling differed in 11 of 12 such launches
    (tools/probes/corun_linear16.py); tests/test_abi.py holds the build to the rule, this holds the two kernels that do packed arithmetic by hand."""
    from patchaugnet_amd import pointops
    from patchaugnet_amd._lib import call, ptr
    from patchaugnet_amd.engine import pack_weights_f16
    from patchaugnet_amd.weights import synthetic_submaps
    clouds = [synthetic_submaps(32, 4096, 90 + i, "street" if i % 2 else "uniform").cuda().squeeze(1).contiguous() for i in range(3)]

    def run(x):
        if victim == "sampling":
            idx, new_xyz = pointops.furthestsampling_gather(x, 1024)
            return idx, new_xyz
        centres = x[:, :1024].contiguous()
        return pointops.knnquery_with_dist(20, x, centres)

    serial = [tuple(t.clone() for t in run(x)) for x in clouds]
    rows, k, n = 131072, 256, 256
    g = torch.Generator().manual_seed(0)
    a = torch.randn(rows, k, generator=g).cuda()
    wt = (torch.randn(k, n, generator=g) / k ** 0.5).cuda().contiguous()
    wp = pack_weights_f16(wt)
    bias = torch.zeros(n, device="cuda")
    out = torch.empty(rows, n, device="cuda")
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    for t in range(9):
        with torch.cuda.stream(sa):
            for _ in range(8):
                call("pa_linear_f16", rows, k, n, ptr(a), k, ptr(wt), ptr(wp), ptr(bias), 1, None, 0, ptr(out), n)
        with torch.cuda.stream(sb):
            got = run(clouds[t % 3])
        torch.cuda.synchronize()
        for u, v in zip(got, serial[t % 3]):
            assert torch.equal(u, v), (victim, t)
