"""PPT-Net grouped self-attention (pa_linear + pa_sa_attention, csrc/attention.hip) and the FC/gating head (pa_fc) on the
MI355X against the CPU restatement of SA_Layer.forward (oracle/models_cpu.sa_layer, pptnet.py:261-282) and the torch modules."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _seed_module(m, seed):
    from patchaugnet_amd.weights import seeded_state_dict
    m.load_state_dict(seeded_state_dict(m.state_dict(), seed=seed))
    return m.eval()


@pytest.mark.parametrize("b,n,c", [(2, 1024, 64), (3, 256, 128), (2, 64, 256), (4, 16, 512), (2, 4, 512), (1, 20, 64), (2, 100, 128),
                                   (1, 333, 64), (2, 37, 256)])
def test_sa_layer_fused_vs_oracle(b, n, c):
    from oracle import models_cpu
    from patchaugnet_amd.backbone import SALayer
    from patchaugnet_amd.engine import _Attn
    sa = _seed_module(SALayer(c, 8), seed=n + c)
    x = torch.randn(b, c, n) * 0.5
    sd = {"s." + k: v for k, v in sa.state_dict().items()}
    with torch.no_grad():
        ref = models_cpu.sa_layer(sd, "s", x, 8)                                          # (B, C, N)
        xm = x.transpose(1, 2).contiguous().view(b * n, c).cuda()
        got = _Attn(sa, xm.device).run(xm, b, n).view(b, n, c).transpose(1, 2).cpu()
        sa = sa.cuda()
        mod = sa(x.cuda()).cpu()                                                          # the module itself: fused HIP path under no_grad
    mod_t = sa.train(False)
    with torch.enable_grad():
        mod_torch = sa(x.cuda()).detach().cpu()                                           # autograd path: torch dense ops
    assert (mod_torch - ref).abs().max().item() <= 2e-5 * max(ref.abs().max().item(), 1.0)
    with torch.no_grad():
        pass
    scale = ref.abs().max().item()
    assert (got - ref).abs().max().item() <= 2e-5 * max(scale, 1.0), (got - ref).abs().max().item()
    assert (mod - ref).abs().max().item() <= 2e-5 * max(scale, 1.0)


@pytest.mark.parametrize("rows,k,n,relu,res", [(1000, 64, 128, 0, False), (77, 259, 256, 1, True), (4096, 512, 1024, 0, False), (33, 20, 16, 1, False),
                                               (512, 512, 1024, 0, False), (512, 512, 512, 1, True), (2048, 256, 512, 0, True), (100, 128, 256, 1, True)])
def test_pa_linear(rows, k, n, relu, res):
    from patchaugnet_amd._lib import call, ptr
    x = torch.randn(rows, k, device="cuda")
    w = torch.randn(n, k, device="cuda") / k ** 0.5
    bias = torch.randn(n, device="cuda")
    r = torch.randn(rows, n, device="cuda") if res else None
    kpad = (k + 3) // 4 * 4
    wt = torch.zeros(kpad, n, device="cuda")
    wt[:k] = w.t()
    out = torch.empty(rows, n, device="cuda")
    call("pa_linear", rows, k, n, ptr(x), k, ptr(wt), None, ptr(bias), relu, ptr(r), n if res else 0, ptr(out), n)
    if n % 64 == 0:                                              # the fragment-major packed weight path must give the same bits
        from patchaugnet_amd.engine import pack_weights
        out2 = torch.empty(rows, n, device="cuda")
        call("pa_linear", rows, k, n, ptr(x), k, ptr(wt), ptr(pack_weights(wt)), ptr(bias), relu, ptr(r), n if res else 0, ptr(out2), n)
        assert torch.equal(out, out2)
    ref = x.double() @ w.double().t() + bias.double()
    if relu:
        ref = ref.clamp_min(0)
    if res:
        ref = ref + r.double()
    assert (out.double() - ref).abs().max().item() <= 2e-5 * max(ref.abs().max().item(), 1.0)


@pytest.mark.parametrize("b,norm", [(2, False), (32, True), (70, False), (130, True)])
def test_ppt_head_fc_and_gating(b, norm):
    """pptnet_origin/models/loupe.py:94-105 -- flat concat -> hidden_weights -> bn2 -> context gating (any batch size)."""
    from patchaugnet_amd import loupe
    from patchaugnet_amd.engine import _FcHead, _Gate
    ks = [1, 4, 16, 64]
    agg = _seed_module(loupe.SpatialPyramidNetVLAD4([256] * 4, [64, 256, 1024, 4096], ks, [256] * 4, gating=True), seed=b).cuda()
    vl = [torch.nn.functional.normalize(torch.randn(b, 256, k, device="cuda"), dim=1) for k in ks]
    with torch.no_grad():
        flat = torch.cat([v.reshape(b, -1) for v in vl], dim=-1)                           # the reference's per-scale C-major flatten
        ref = agg.bn2(torch.matmul(flat, agg.hidden_weights))
        ref = agg.context_gating(ref)
        if norm:
            ref = torch.nn.functional.normalize(ref)
        rows = torch.cat(vl, dim=-1).transpose(1, 2).contiguous()                          # (B, 85, 256) rows of the VLAD kernel
        h = _FcHead(agg.hidden_weights, agg.bn2, ks, per_scale=True, l2=0, device=flat.device).run(rows)
        got = _Gate(agg.context_gating, l2=1 if norm else 0, device=flat.device).run(h)
    assert (got - ref).abs().max().item() <= 3e-5 * max(ref.abs().max().item(), 1.0)


@pytest.mark.parametrize("b,kdim,nout", [(32, 21504, 256), (5, 1000, 80), (70, 516, 128), (3, 130, 16)])
def test_pa_fc_both_operand_paths(b, kdim, nout):
    """pa_fc (split-K FC + BatchNorm fold + optional L2 normalise): the 16-byte operand path (nout % 64 == 0, kdim % 4 == 0) and the 4-byte
    fallback (any nout % 16 == 0, ragged kdim) against fp64 torch."""
    from patchaugnet_amd import _lib
    from patchaugnet_amd._lib import call, ptr
    g = torch.Generator(device="cuda").manual_seed(kdim + nout)
    y = torch.randn(b, kdim, device="cuda", generator=g)
    wt = torch.randn(kdim, nout, device="cuda", generator=g) * 0.05
    bias = torch.randn(nout, device="cuda", generator=g)
    scale = torch.rand(nout, device="cuda", generator=g) + 0.5
    shift = torch.randn(nout, device="cuda", generator=g)
    scratch = torch.empty(_lib.lib().pa_fc_scratch_floats(b, kdim, nout), device="cuda")
    for l2 in (0, 1):
        out = torch.empty(b, nout, device="cuda")
        call("pa_fc", b, kdim, nout, ptr(y), ptr(wt), ptr(bias), ptr(scale), ptr(shift), l2, None, ptr(scratch), ptr(out))
        ref = (y.double() @ wt.double() + bias.double()) * scale.double() + shift.double()
        if l2:
            ref = torch.nn.functional.normalize(ref, dim=1)
        err = ((out.double() - ref).abs().max() / ref.abs().max()).item()
        assert err <= 2e-5, (l2, err)
