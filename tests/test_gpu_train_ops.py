"""Training-mode dense path (csrc/train_gemm.hip + patchaugnet_amd/train_ops.py) on the MI355X.

1. the two GEMM kernels with every operand transform, against fp64 torch statements of include/patchaugnet_hip.h's definitions
   (ragged shapes: nothing a multiple of a tile);
2. the autograd layer: SharedMLP (pt_util.py:16-41 in train() mode) with and without the fused max-pool, PointNetDecoder
   (pointnet_autoencoder.py:85-111), NetVLADBase / APFA / context gating (loupe.py), the grouped self-attention of PPT-Net
   (pptnet.py:246-282) -- outputs, input / parameter gradients and BatchNorm running statistics, in train() AND in eval() mode, against
   torch autograd of a float64 CPU copy of the SAME module (the modules' CPU form is plain torch; on the device they have no torch-dense
   branch to switch to), tolerance 2e-4 relative to the tensor's scale.
"""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = a.double(), b.double()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def _rel_robust(a, b, tol):
    """Comparison used ONLY when the torch run saw a ReLU input within rounding distance of zero (see _compare): that activation gets
    the mask of one side only, which moves the gradients of that one point by O(1) -- half of the elements within 20 tol of the tensor's
    scale and the whole tensor within 1e-2 in relative L2."""
    a, b = a.double().flatten(), b.double().flatten()
    d = (a - b).abs()
    scale = b.abs().max().clamp_min(1e-30)
    q = torch.quantile(d[:: max(d.numel() // 1000000, 1)], 0.5)
    e = ((q / scale).item(), (d.norm() / b.norm().clamp_min(1e-30)).item())
    return e[0] <= 20 * tol and e[1] <= 1e-2, e


def _p_block(nch, g):
    p = torch.randn(7, nch, generator=g)
    p[3] = p[3].abs() + 0.5
    return p


def _tf(mode, gv, yv, p):
    """operand transform of train_gemm.hip per channel (rows of gv / yv); p (7, nch) fp64"""
    c = [p[j][:, None] for j in range(7)]
    if mode == 0:
        return gv
    if mode == 1:
        return torch.relu(gv * c[0] + c[1])
    z = yv * c[0] + c[1]
    gm = torch.where(z > 0, gv, torch.zeros_like(gv)) if mode == 2 else gv
    xhat = (yv - c[2]) * c[3]
    return (gm - c[4] - xhat * c[5]) * c[6]


@pytest.mark.parametrize("M,N,K,batch", [(70, 300, 37, 3), (128, 128, 16, 1), (200, 1000, 259, 2), (18, 515, 64, 1), (33, 7, 5, 2)])
@pytest.mark.parametrize("a_kcontig", [True, False])
@pytest.mark.parametrize("bmode", [0, 1, 2, 3])
def test_tgemm_nn(M, N, K, batch, a_kcontig, bmode):
    from patchaugnet_amd import train_ops as T
    g = torch.Generator().manual_seed(M * 131 + N * 7 + K + bmode)
    shared = (M + bmode) % 2 == 0
    A = torch.randn((1 if shared else batch, M, K) if a_kcontig else (1 if shared else batch, K, M), generator=g)
    Bm = torch.randn(batch, K, N, generator=g)
    aux = torch.randn(batch, K, N, generator=g)
    p = _p_block(K, g)
    bias = torch.randn(M, generator=g)
    C0 = torch.randn(batch, M, N, generator=g)
    for beta, use_bias, act, use_stats in ((0, False, 0, True), (1, True, 0, False), (0, True, 1, True)):
        C = C0.clone().cuda()
        stats = torch.zeros(T.STAT_SLOTS, 2, M, dtype=torch.float64, device="cuda")
        Ad, Bd, auxd, pd, biasd = A.cuda(), Bm.cuda(), aux.cuda(), p.cuda().contiguous(), bias.cuda()
        T.tgemm_nn(batch, M, N, K, Ad, 0 if shared else M * K, K if a_kcontig else M, a_kcontig, Bd, K * N, N, C, M * N, N, bmode=bmode,
                   baux=auxd if bmode >= 2 else None, bp=pd if bmode else None, beta=beta, bias=biasd if use_bias else None, act=act,
                   stats=stats if use_stats else None)
        torch.cuda.synchronize()
        A64 = A.double() if a_kcontig else A.double().transpose(1, 2)
        fB = torch.stack([_tf(bmode, Bm[b].double(), aux[b].double(), p.double()) for b in range(batch)])
        ref = torch.matmul(A64, fB)
        if use_bias:
            ref = ref + bias.double()[None, :, None]
        if act == 1:
            ref = torch.tanh(ref)
        if beta:
            ref = ref + C0.double()
        scale = ref.abs().max().item()
        err = (C.cpu().double() - ref).abs().max().item()
        assert err <= 2e-5 * max(scale, 1.0) + 1e-6 * K, (beta, use_bias, act, err, scale)
        if use_stats:
            v = C.cpu().double()
            s = torch.stack([v.sum((0, 2)), (v * v).sum((0, 2))])
            assert torch.allclose(stats.sum(0).cpu(), s, rtol=1e-5, atol=1e-4 * max(scale, 1.0)), (stats.sum(0).cpu() - s).abs().max()


def test_tgemm_nn_128_row_tile_variant_in_a_subprocess():
    """The 128-row tiles of pa_tgemm_nn are an A/B knob now (PA_TGEMM_BIG_MIN, read once per process): run the GEMM cases with it forced on."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if os.environ.get("PA_TGEMM_BIG_MIN"):
        pytest.skip("already inside the forced-variant run")
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_train_ops.py"), "-q", "-m", "gpu", "-k", "test_tgemm_nn and not subprocess",
                          "-p", "no:cacheprovider"], capture_output=True, text=True, timeout=900, cwd=root, env=dict(os.environ, PA_TGEMM_BIG_MIN="1"))
    assert out.returncode == 0 and " passed" in out.stdout, out.stdout[-2000:] + out.stderr[-1000:]


@pytest.mark.parametrize("M,N,K,batch", [(70, 37, 300, 3), (64, 64, 128, 1), (200, 259, 1001, 2), (18, 256, 21504, 1), (5, 3, 7, 2),
                                        (32, 6, 20480, 18), (64, 64, 90001, 3)])      # small outputs over long contractions: the replicated path (pa_tgemm_kk_rep)
@pytest.mark.parametrize("amode", [0, 2, 3])
@pytest.mark.parametrize("bmode", [0, 1])
def test_tgemm_kk(M, N, K, batch, amode, bmode):
    from patchaugnet_amd import train_ops as T
    g = torch.Generator().manual_seed(M * 131 + N * 7 + K + amode * 3 + bmode)
    A = torch.randn(batch, M, K, generator=g)
    aux = torch.randn(batch, M, K, generator=g)
    Bm = torch.randn(batch, N, K, generator=g)
    pa, pb = _p_block(M, g), _p_block(N, g)
    fA = torch.stack([_tf(amode, A[b].double(), aux[b].double(), pa.double()) for b in range(batch)])
    fB = torch.stack([_tf(bmode, Bm[b].double(), None, pb.double()) for b in range(batch)])
    per = torch.matmul(fA, fB.transpose(1, 2))
    for per_batch in (0, 1):
        C0 = torch.randn((batch, M, N) if per_batch else (M, N), generator=g)
        C = C0.clone().cuda()
        T.tgemm_kk(batch, M, N, K, A.cuda(), M * K, K, Bm.cuda(), N * K, K, C, M * N if per_batch else 0, N, amode=amode,
                   aaux=aux.cuda() if amode else None, ap=pa.cuda() if amode else None, bmode=bmode, bp=pb.cuda() if bmode else None,
                   per_batch=per_batch)
        torch.cuda.synchronize()
        ref = C0.double() + (per if per_batch else per.sum(0))
        scale = ref.abs().max().item()
        err = (C.cpu().double() - ref).abs().max().item()
        assert err <= 2e-5 * max(scale, 1.0) + 2e-7 * K, (per_batch, err, scale)


@pytest.mark.parametrize("B,C,Pout,pool,relu", [(18, 64, 1024, 20, 1), (3, 256, 128, 20, 1), (2, 5, 37, 3, 0), (2, 32, 300, 16, 1)])
def test_maxpool_backward_with_the_pooled_layers_bn_backward_sums(B, C, Pout, pool, relu):
    """pa_maxpool_bwd_bnred (the scatter of the pooled gradient + the pooled layer's BatchNorm-backward sums from the Pout pooled gradients and the
    raw outputs at their arg-max positions) against the pair it replaces: pa_maxpool_bwd, then pa_bn_bwd_reduce over the dense (gradient, raw
    output) tensors -- the scattered gradient bit for bit, the sums to fp64 summation order."""
    from patchaugnet_amd._lib import call, ptr
    g = torch.Generator().manual_seed(B * 100 + C + Pout)
    gp = torch.randn(B, C, Pout, generator=g).cuda()
    arg = torch.randint(0, pool, (B, C, Pout), generator=g).to(torch.int8).cuda()
    y = torch.randn(B, C, Pout * pool, generator=g).cuda()
    p = _p_block(C, g).cuda().contiguous()
    full0 = torch.empty(B, C, Pout * pool, device="cuda")
    call("pa_maxpool_bwd", B * C, Pout, pool, ptr(gp), ptr(arg), ptr(full0))
    s0 = torch.zeros(2 * C, dtype=torch.float64, device="cuda")
    call("pa_bn_bwd_reduce", B, C, Pout * pool, ptr(full0), ptr(y), ptr(p), relu, ptr(s0), 0)
    full1 = torch.full((B, C, Pout * pool), 9.0, device="cuda")
    s1 = torch.zeros(2 * C, dtype=torch.float64, device="cuda")
    call("pa_maxpool_bwd_bnred", B, C, Pout, pool, ptr(gp), ptr(arg), ptr(full1), ptr(y), ptr(p), relu, ptr(s1))
    torch.cuda.synchronize()
    assert torch.equal(full0, full1)
    assert torch.allclose(s0, s1, rtol=1e-5, atol=1e-4 * (B * Pout) ** 0.5), (s0 - s1).abs().max()


def _grads(mod, x, fn, gout_seed=3):
    for p in mod.parameters():
        p.grad = None
    xx = x.clone().requires_grad_(True)
    out = fn(mod, xx)
    go = torch.randn(out.shape, generator=torch.Generator().manual_seed(gout_seed)).to(out.device, out.dtype)
    out.backward(go)
    gr = {k: p.grad.clone() for k, p in mod.named_parameters() if p.grad is not None}
    return out.detach(), xx.grad.clone(), gr, {k: v.clone() for k, v in mod.state_dict().items() if "running" in k or "tracked" in k}


def _randomize_running_stats(mod, seed=5):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for m in mod.modules():
            if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
                m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.3)
                m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) + 0.5)


def _compare(mod, x, fn, tol=2e-4, skip=(), train=True):
    """fn(module, input) on the device module (HIP kernels) against the same call on a float64 CPU copy (plain torch autograd)."""
    mod.train(train)
    ref = copy.deepcopy(mod).cpu().double()
    o1, dx1, g1, s1 = _grads(mod, x, fn)
    near = []
    hooks = [m.register_forward_hook(lambda _m, _i, o: near.append(o.detach().abs().min().item()))
             for m in ref.modules() if isinstance(m, torch.nn.modules.batchnorm._BatchNorm)]
    o0, dx0, g0, s0 = _grads(ref, x.detach().cpu().double(), fn)
    for h in hooks:
        h.remove()
    o1, dx1 = o1.cpu(), dx1.cpu()
    g1 = {k: v.cpu() for k, v in g1.items()}
    s1 = {k: v.cpu() for k, v in s1.items()}
    flip = min(near, default=1.0) < 2e-6          # a ReLU input at rounding distance from zero: its mask may differ between the two runs
    def close(a, b):
        return _rel_robust(a, b, tol) if flip else (_rel(a, b) <= tol, _rel(a, b))
    assert _rel(o1, o0) <= tol, ("out", _rel(o1, o0))
    ok, e = close(dx1, dx0)
    assert ok, ("dx", e, flip)
    assert set(g1) == set(g0), set(g1) ^ set(g0)
    for k in g0:
        if any(s in k for s in skip):
            assert g1[k].abs().max().item() <= 1e-4 * max(1.0, g0[k].abs().max().item()) + 1e-5, k   # mathematically zero on both sides
            continue
        ok, e = close(g1[k], g0[k])
        assert ok, (k, e, flip)
    for k in s0:
        assert torch.allclose(s1[k].double(), s0[k].double(), rtol=1e-4, atol=1e-5), k


@pytest.mark.parametrize("spec,shape", [([6, 32, 32, 64], (3, 6, 50, 20)), ([67, 64, 64, 256], (2, 67, 37, 20)), ([259, 256, 256], (2, 259, 300, 1)),
                                        ([768, 256, 256], (3, 768, 16, 1)),
                                        ([6, 32], (2, 6, 300, 13)), ([6, 16, 32], (2, 6, 700, 20)),      # pooled outputs beyond one 256-wide block; nsample % 4 != 0
                                        ([8, 512], (144, 8, 4, 20))])     # 144 clouds x 512 channels = 73 728 (cloud, channel) rows > 65 535
def test_shared_mlp_train_matches_torch_autograd(spec, shape):
    from patchaugnet_amd.pt_util import SharedMLP
    torch.manual_seed(1)
    m = SharedMLP(spec, bn=True).cuda().train()
    with torch.no_grad():
        for l in m:
            l.bn.bn.weight.uniform_(0.5, 1.5)
            l.bn.bn.bias.normal_(0, 0.2)
    x = torch.randn(shape, device="cuda")
    _compare(m, x, lambda mod, t: mod(t))
    if shape[3] > 1:
        _compare(m, x, lambda mod, t: mod.forward_maxpool(t))
    # eval() with autograd: the running statistics, same kernels (no torch.matmul / MIOpen route on the device)
    _randomize_running_stats(m)
    _compare(m, x, lambda mod, t: mod(t), train=False)
    if shape[3] > 1:
        _compare(m, x, lambda mod, t: mod.forward_maxpool(t), train=False)


def test_decoder_train_matches_torch_autograd():
    from patchaugnet_amd.patch_aug_net import PointNetDecoder
    torch.manual_seed(2)
    d = PointNetDecoder(embedding_size=256, num_points=20).cuda().train()
    x = torch.nn.functional.normalize(torch.randn(1024, 256, device="cuda"))
    _compare(d, x, lambda mod, t: mod(t), skip=("fc1.bias", "fc2.bias"))
    # R related clouds in one set of launches (every cloud its own BatchNorm batch) == the decoder called once per cloud, in order
    x3 = torch.nn.functional.normalize(torch.randn(3, 256, 1024, device="cuda"), dim=1)

    def run(mod, t):
        if t.is_cuda:
            return mod.forward_cm(t)
        return torch.stack([mod(t[r].t()) for r in range(t.shape[0])])
    _compare(d, x3, run, skip=("fc1.bias", "fc2.bias"))
    # eval(): running statistics; the biases in front of the BatchNorms now HAVE a gradient (no mean subtraction)
    _randomize_running_stats(d)
    _compare(d, x, lambda mod, t: mod(t), train=False)
    _compare(d, x3, run, train=False)


@pytest.mark.parametrize("C,N,K", [(256, 4096, 64), (256, 1024, 16), (256, 128, 4), (64, 100, 5)])
def test_netvlad_train_matches_torch_autograd(C, N, K):
    from patchaugnet_amd.loupe import NetVLADBase
    torch.manual_seed(3)
    v = NetVLADBase(C, N, K, C).cuda().train()
    x = torch.randn(3, C, N, 1, device="cuda")
    _compare(v, x, lambda mod, t: mod(t))
    _randomize_running_stats(v)
    _compare(v, x, lambda mod, t: mod(t), train=False)


def test_heads_train_match_torch_autograd():
    from patchaugnet_amd.loupe import AdaptiveFeatureAggregator, GatingContext, SpatialPyramidNetVLAD
    torch.manual_seed(4)
    afa = AdaptiveFeatureAggregator(256, 84, 256).cuda().train()
    x = torch.randn(18, 256, 84, device="cuda")
    _compare(afa, x, lambda mod, t: mod(t), skip=("fc.bias",))
    _randomize_running_stats(afa)
    _compare(afa, x, lambda mod, t: mod(t), train=False)
    gate = GatingContext(256).cuda().train()
    x = torch.randn(18, 256, device="cuda")
    _compare(gate, x, lambda mod, t: mod(t))
    _randomize_running_stats(gate)
    _compare(gate, x, lambda mod, t: mod(t), train=False)
    for agg in (0, 2, 3):
        sp = SpatialPyramidNetVLAD([256, 256, 256], [64, 256, 512], [4, 8, 16], [256, 256, 256], gating=True, aggregation_type=agg).cuda().train()
        feats = [torch.randn(4, 256, n, 1, device="cuda") for n in (64, 256, 512)]

        def run(mod, t, feats=feats):
            return mod([t] + [f.to(t.device, t.dtype) for f in feats[1:]])
        _compare(sp, feats[0], run, tol=5e-4, skip=("fc.bias",))
        _randomize_running_stats(sp)
        _compare(sp, feats[0], run, tol=5e-4, train=False)


@pytest.mark.parametrize("C,N,B", [(64, 1024, 2), (128, 256, 3), (256, 64, 2), (512, 16, 4), (64, 100, 1)])
def test_grouped_self_attention_autograd_matches_torch(C, N, B):
    """SA_Layer (pptnet.py:246-282) in train() and in eval()-with-autograd on the MI355X: csrc/train_gemm.hip GEMMs + csrc/attention_train.hip
    soft-max / column re-normalisation, forward and backward, against the module's CPU form in float64."""
    from patchaugnet_amd.backbone import SALayer
    torch.manual_seed(6)
    sa = SALayer(C, 8).cuda().train()
    with torch.no_grad():
        sa.k_conv.weight.mul_(0.5)                  # keep the soft-max away from one-hot rows (O(C) logits otherwise)
        sa.after_norm.weight.uniform_(0.5, 1.5)
        sa.after_norm.bias.normal_(0, 0.2)
    x = torch.randn(B, C, N, device="cuda") * 0.5
    # train(): trans_conv.bias sits in front of a BatchNorm, and v_conv.bias enters x_r as bias x (column sums of A = 1), a per-channel constant
    # along the points that the same BatchNorm removes: both gradients are mathematically zero
    _compare(sa, x, lambda mod, t: mod(t), tol=5e-4, skip=("trans_conv.bias", "v_conv.bias"))
    _randomize_running_stats(sa)
    _compare(sa, x, lambda mod, t: mod(t), tol=5e-4, train=False)
    # eval() under no_grad is the fused two-pass kernel of csrc/attention.hip: same function
    sa.eval()
    with torch.no_grad():
        fused = sa(x)
    ref = copy.deepcopy(sa).cpu().double()(x.cpu().double())
    assert _rel(fused.cpu(), ref) <= 5e-4


def test_attention_softmax_renorm_kernels_against_fp64():
    """pa_attn_softmax_renorm / _backward (csrc/attention_train.hip) on ragged sizes, against torch fp64 autograd of pptnet.py:276-277."""
    from patchaugnet_amd import train_ops
    g = torch.Generator().manual_seed(9)
    for b, n in [(2, 1024), (3, 100), (1, 257), (4, 16)]:
        e = (torch.randn(b, n, n, generator=g) * 2).cuda()
        e64 = e.cpu().double().requires_grad_(True)
        p = torch.softmax(e64, dim=-1)
        a64 = p / (1e-9 + p.sum(dim=1, keepdim=True))
        go = torch.randn(b, n, n, generator=g)
        a64.backward(go.double())
        ed = e.clone().requires_grad_(True)
        a = train_ops._SoftmaxRenorm.apply(ed * 1.0)
        a.backward(go.cuda())
        assert _rel(a.detach().cpu(), a64.detach()) <= 1e-5
        assert _rel(ed.grad.cpu(), e64.grad) <= 2e-5


def _library_kernels(step):
    step()
    with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA]) as prof:
        step()
        torch.cuda.synchronize()
    names = [e.key for e in prof.key_averages()]
    lib = [n for n in names if any(s in n for s in ("Cijk", "rocblas", "miopen", "MIOpen", "hipblas", "gemm_kernel", "batch_norm", "BatchNorm",
                                                      "naive_conv", "igemm", "Conv", "conv"))]
    return names, lib


@pytest.mark.parametrize("family,mode", [("patch_aug_net", "train"), ("patch_aug_net", "eval_grad"), ("patch_aug_net", "eval_nograd_module"),
                                         ("pptnet", "train"), ("pptnet", "eval_grad")])
def test_module_path_has_no_library_gemm(family, mode):
    """Forward (+ backward) of the module path on the MI355X -- train(), eval() with autograd, eval() under no_grad with the fused engine switched
    off; PatchAugNet and PPT-Net (grouped self-attention) -- the profiler sees no rocBLAS / MIOpen / hipBLASLt kernel, only the HIP GEMMs."""
    from patchaugnet_amd import configs, patch_aug_net, pptnet
    from patchaugnet_amd.weights import seeded_state_dict
    if family == "patch_aug_net":
        m = patch_aug_net.Network(param=configs.patch_aug_net_config(), use_a2a_recon=True, use_l2_norm=True)
    else:
        m = pptnet.Network(param=configs.pptnet_config(), use_normalize=True)
    m.load_state_dict(seeded_state_dict(m.state_dict()))
    m = m.cuda()
    m.train(mode == "train")
    grad = mode != "eval_nograd_module"
    x = (torch.rand(3, 1, 4096, 3, generator=torch.Generator().manual_seed(0)) * 2 - 1).cuda().requires_grad_(grad)
    nn_dict = {(0, 1): np.zeros((4, 1), np.int64)} if family == "patch_aug_net" else None

    def step():
        with torch.set_grad_enabled(grad):
            if nn_dict is not None:
                (desc, recon), _, _ = m(x, nn_dict)
                loss = desc.square().sum() + sum(r.square().mean() for r in recon["reconstructed_patches"])
            else:
                out = m(x, use_engine=False) if family == "patch_aug_net" else m(x)
                desc = out[0] if isinstance(out, tuple) else out
                loss = desc.square().sum()
            if grad:
                loss.backward()
    names, lib = _library_kernels(step)
    assert any("tgemm_nn_kernel" in n for n in names), names[:40]
    assert not grad or any("tgemm_kk_kernel" in n for n in names), names[:40]
    assert not lib, lib
    if family == "pptnet":
        assert any("at_row_softmax_kernel" in n for n in names) and (not grad or any("at_bwd_row_kernel" in n for n in names)), names[:60]


def test_graphed_training_step_equals_the_eager_step():
    """train.GraphedTrainer (forward + losses + backward + optimizer step in ONE hipGraph) against train.training_step on the same
    weights, inputs and kNN permutations.  SGD for the three-step trajectory (an update proportional to the gradient keeps the atomics'
    last-bit noise small; Adam's g / sqrt(v) turns it into sign flips on near-zero gradients), Adam(capturable) for one step."""
    import copy
    from patchaugnet_amd import configs, patch_aug_net
    from patchaugnet_amd.train import DEFAULTS, GraphedTrainer, training_step
    from patchaugnet_amd.weights import seeded_state_dict
    n = 1024
    cfg = configs.scaled_config(configs.patch_aug_net_config(), n)
    base = patch_aug_net.Network(param=cfg, use_a2a_recon=True, use_l2_norm=True)
    base.load_state_dict(seeded_state_dict(base.state_dict()))
    base = base.cuda()
    g = torch.Generator().manual_seed(3)
    batches = [tuple((torch.rand(1, k, n, 3, generator=g) * 2 - 1).cuda() for k in (1, 2, 4, 1)) for _ in range(3)]
    nn_dict = {(0, 1): None, (0, 2): None}
    args = dict(DEFAULTS, TRAIN_NEGATIVES_PER_QUERY=4)
    for make, steps in ((lambda ps, cap: torch.optim.SGD(ps, lr=2e-3), 3), (lambda ps, cap: torch.optim.Adam(ps, lr=1e-4, capturable=cap), 1)):
        m0, m1 = copy.deepcopy(base), copy.deepcopy(base)
        o0, o1 = make(m0.parameters(), False), make(m1.parameters(), True)
        tr = GraphedTrainer(m1, o1, *batches[0], nn_dict, num_points=n, args=args, warmup=2)
        m1.load_state_dict(base.state_dict())               # the warm-up steps moved m1: start both from the same point again ...
        for st in o1.state.values():                        # ... and the optimizer state IN PLACE (the graph holds these tensors' addresses)
            for v in st.values():
                if torch.is_tensor(v):
                    v.zero_()
        for i, b in enumerate(batches[:steps]):
            torch.manual_seed(11 + i)
            e = training_step(m0, o0, *b, nn_dict=nn_dict, num_points=n, args=args)
            torch.manual_seed(11 + i)
            gl = {k: float(v) for k, v in tr.step(*b).items()}
            for k in e:
                assert abs(e[k] - gl[k]) <= (2e-5 if i == 0 else 1e-2) * max(1.0, abs(e[k])), (i, k, e[k], gl[k])
        sd0, sd1 = m0.state_dict(), m1.state_dict()
        # the two trajectories share every kernel but not the order of their fp32 atomics: per tensor within 1e-3 in relative L2
        # (tensors of norm < 1: absolute), worst single element within 3e-2 of the tensor's largest
        for k in sd0:
            a, c = sd0[k].double(), sd1[k].double()
            assert ((a - c).norm() / a.norm().clamp_min(1.0)).item() <= 5e-3, k
            assert ((a - c).abs().max() / a.abs().max().clamp_min(1e-2)).item() <= 3e-2, k
        tr.close()


def test_graphed_training_step_with_eager_launches_between_replays():
    """Two replays of the captured step at learning rate 0 (same weights, same inputs, same kNN permutation) must produce the same parameter
    gradients -- also when other work (allocations, a fill launch, a device-to-host read: what any real loop does between steps) runs on the
    stream between them.  Until round 5 the patch-Chamfer backward zeroed its accumulation targets with hipMemsetAsync; as memset NODES of the
    captured graph those were not reliably ordered in front of the scatter kernels, and a fraction of the replays accumulated onto stale memory:
    |gradient| of 1e20 .. inf through the whole reconstruction branch (decoder, coarse FP levels, every SA level)."""
    import copy
    from patchaugnet_amd import configs, patch_aug_net
    from patchaugnet_amd.train import DEFAULTS, GraphedTrainer
    from patchaugnet_amd.weights import seeded_state_dict
    n = 1024
    cfg = configs.scaled_config(configs.patch_aug_net_config(), n)
    m = patch_aug_net.Network(param=cfg, use_a2a_recon=True, use_l2_norm=True)
    m.load_state_dict(seeded_state_dict(m.state_dict()))
    m = m.cuda()
    g = torch.Generator().manual_seed(5)
    batch = tuple((torch.rand(1, k, n, 3, generator=g) * 2 - 1).cuda() for k in (1, 2, 4, 1))
    opt = torch.optim.SGD(m.parameters(), lr=0.0)
    tr = GraphedTrainer(m, opt, *batch, {(0, 1): None, (0, 2): None}, num_points=n, args=dict(DEFAULTS, TRAIN_NEGATIVES_PER_QUERY=4), warmup=2)
    scratch = torch.empty(1 << 20, device="cuda")
    torch.manual_seed(1)
    tr.step(*batch)
    torch.cuda.synchronize()
    ref = {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}
    for rep in range(6):
        scratch.fill_(float(rep))                                               # an eager launch ...
        junk = [torch.full((1 << 20,), float("nan"), device="cuda") for _ in range(8)]      # ... allocations whose content must never be seen ...
        float(scratch[0])                                                       # ... and a read-back between replays
        del junk
        torch.manual_seed(1)
        tr.step(*batch)
        torch.cuda.synchronize()
        for k, p in m.named_parameters():
            if k in ref:
                a, b = ref[k], p.grad
                assert torch.isfinite(b).all(), (rep, k)
                assert (a - b).norm().item() <= 2e-2 * max(a.norm().item(), 1e-3), (rep, k, a.norm().item(), (a - b).norm().item())
    tr.close()


def test_graphed_trainer_and_changed_optimizer_hyperparameters():
    """A captured step replays its launch constants.  With patchaugnet_amd.optim.Adam (learning rate in a device scalar) a scheduler's new
    learning rate takes effect on the next replay: after lr -> 0 a step leaves the weights where they were.  With torch's capturable Adam
    (float lr baked into the captured launches) the same change raises instead of being ignored, and so does a changed beta on either."""
    import copy
    from patchaugnet_amd import configs, patch_aug_net
    from patchaugnet_amd.optim import Adam
    from patchaugnet_amd.train import DEFAULTS, GraphedTrainer
    from patchaugnet_amd.weights import seeded_state_dict
    n = 1024
    cfg = configs.scaled_config(configs.patch_aug_net_config(), n)
    base = patch_aug_net.Network(param=cfg, use_a2a_recon=True, use_l2_norm=True)
    base.load_state_dict(seeded_state_dict(base.state_dict()))
    base = base.cuda()
    g = torch.Generator().manual_seed(5)
    batch = tuple((torch.rand(1, k, n, 3, generator=g) * 2 - 1).cuda() for k in (1, 2, 4, 1))
    nn_dict = {(0, 1): None, (0, 2): None}
    args = dict(DEFAULTS, TRAIN_NEGATIVES_PER_QUERY=4)
    m = copy.deepcopy(base)
    opt = Adam(m.parameters(), lr=1e-3)
    tr = GraphedTrainer(m, opt, *batch, nn_dict, num_points=n, args=args, warmup=2)
    tr.step(*batch)
    torch.cuda.synchronize()
    w = {k: v.clone() for k, v in m.state_dict().items() if v.dtype == torch.float32 and "running" not in k}
    tr.step(*batch)
    torch.cuda.synchronize()
    moved = max((m.state_dict()[k] - w[k]).abs().max().item() for k in w)
    assert moved > 1e-5                                                         # lr = 1e-3: the step moves the weights
    opt.param_groups[0]["lr"] = 0.0                                             # what a scheduler does
    w = {k: v.clone() for k, v in m.state_dict().items() if k in w}
    tr.step(*batch)
    torch.cuda.synchronize()
    diffs = {k: (m.state_dict()[k] - w[k]).abs().max().item() for k in w}
    assert all(d == 0.0 for d in diffs.values()), {k: d for k, d in diffs.items() if d != 0.0}     # the replay used the new learning rate
    opt.param_groups[0]["betas"] = (0.5, 0.999)
    with pytest.raises(RuntimeError):
        tr.step(*batch)
    tr.close()
    m2 = copy.deepcopy(base)
    opt2 = torch.optim.Adam(m2.parameters(), lr=1e-3, capturable=True)
    tr2 = GraphedTrainer(m2, opt2, *batch, nn_dict, num_points=n, args=args, warmup=2)
    tr2.step(*batch)
    opt2.param_groups[0]["lr"] = 1e-4
    with pytest.raises(RuntimeError):
        tr2.step(*batch)
    tr2.close()


def test_graphed_trainer_with_geometry_prefetch_equals_the_eager_steps():
    """GraphedTrainer(prefetch=True): sampling / neighbour search / 3-NN of the NEXT batch replayed on a side stream under the current step,
    two buffer sets alternating.  Three SGD steps on three different batches against train.training_step from the same weights; the host
    RNG is seeded once per run, so both draw the same sequence of kNN permutations (the trainer draws batch i + 1's during call i).  Also: a
    call whose batch was NOT announced (the first one, and one after a skipped announcement) computes its geometry inline."""
    import copy
    from patchaugnet_amd import configs, patch_aug_net
    from patchaugnet_amd.train import DEFAULTS, GraphedTrainer, training_step
    from patchaugnet_amd.weights import seeded_state_dict
    n = 1024
    cfg = configs.scaled_config(configs.patch_aug_net_config(), n)
    base = patch_aug_net.Network(param=cfg, use_a2a_recon=True, use_l2_norm=True)
    base.load_state_dict(seeded_state_dict(base.state_dict()))
    base = base.cuda()
    g = torch.Generator().manual_seed(4)
    # three steps: two runs of the EAGER step already differ by ~2 % in the fourth step's hinge loss (order of the fp32 atomics)
    batches = [tuple((torch.rand(1, k, n, 3, generator=g) * 2 - 1).cuda() for k in (1, 2, 4, 1)) for _ in range(3)]
    nn_dict = {(0, 1): None, (0, 2): None}
    args = dict(DEFAULTS, TRAIN_NEGATIVES_PER_QUERY=4)
    m0, m1 = copy.deepcopy(base), copy.deepcopy(base)
    o0, o1 = torch.optim.SGD(m0.parameters(), lr=2e-3), torch.optim.SGD(m1.parameters(), lr=2e-3)
    tr = GraphedTrainer(m1, o1, *batches[0], nn_dict, num_points=n, args=args, warmup=2, prefetch=True)
    assert tr.prefetch and len(tr.graphs) == 2 and len(tr.geo_graphs) == 2
    m1.load_state_dict(base.state_dict())
    torch.manual_seed(21)
    eager = [training_step(m0, o0, *b, nn_dict=nn_dict, num_points=n, args=args) for b in batches]
    torch.manual_seed(21)
    got = []
    for i, b in enumerate(batches):
        # call 0 computes its geometry inline and announces batch 1; call 1 runs on the prefetched set and announces nothing; call 2 must
        # notice that and compute its geometry itself (on the set whose stale prefetch it first waits for)
        nb = batches[i + 1] if i == 0 else None
        got.append({k: float(v) for k, v in tr.step(*b, next_batch=nb).items()})
    torch.cuda.synchronize()
    for i, (e, gl) in enumerate(zip(eager, got)):
        for k in e:
            assert abs(e[k] - gl[k]) <= (2e-5 if i == 0 else 1e-2) * max(1.0, abs(e[k])), (i, k, e[k], gl[k])
    sd0, sd1 = m0.state_dict(), m1.state_dict()
    for k in sd0:
        a, c = sd0[k].double(), sd1[k].double()
        # (two runs of the same three steps differ by the order of their fp32 atomics -- split-K weight gradients, the folded level's skip-column
        # gradient -- and a ReLU input at rounding distance from zero then takes the other branch: observed up to 5.4e-3 between eager and replayed)
        assert ((a - c).norm() / a.norm().clamp_min(1.0)).item() <= 1e-2, k
        assert ((a - c).abs().max() / a.abs().max().clamp_min(1e-2)).item() <= 5e-2, k
    tr.close()


def test_linear_rows_single_row_leaves_the_bias_parameter_alone():
    """nn.Linear on ONE row (a batch of one cloud through the APFA head): the split-K kernel accumulates into its output, which must start
    from a copy of the bias -- bias.expand(1, O).contiguous() is a view of the parameter itself (found by the head fuzz family)."""
    from patchaugnet_amd import train_ops
    torch.manual_seed(0)
    for rows in (1, 2):
        x = torch.randn(rows, 256 * 93, device="cuda")
        W = torch.randn(256, 256 * 93, device="cuda") * 0.01
        bias = torch.randn(256, device="cuda")
        keep = bias.clone()
        for _ in range(2):
            y = train_ops.linear_rows(x, W, bias)
            assert torch.equal(bias, keep)
            assert _rel(y, x.double() @ W.double().t() + keep.double()) <= 1e-5


def _cm_switch(on):
    import ctypes
    from patchaugnet_amd import _lib
    lib = _lib.lib()
    lib.pa_tgemm_cm_enable.argtypes, lib.pa_tgemm_cm_enable.restype = [ctypes.c_int], None
    lib.pa_tgemm_cm_enable(on)


@pytest.mark.parametrize("a_kcontig,bmode,use_stats", [(True, 0, True), (True, 1, True), (False, 0, False), (False, 2, False), (False, 3, False), (True, 2, False),
                                                       (False, 1, True)])
@pytest.mark.parametrize("B,M,N,K,bias", [(3, 64, 128, 64, False), (2, 256, 320, 256, True), (1, 128, 64, 128, False), (5, 192, 192, 256, False),
                                          (18, 256, 1024, 256, False), (2, 512, 64, 64, True), (3, 64, 96, 64, False), (7, 128, 416, 128, True),
                                          (4, 32, 2048, 64, False), (3, 96, 128, 32, True), (2, 160, 64, 32, False), (5, 64, 4096, 32, False)])
def test_lds_resident_weights_gemm_against_float64_and_the_lds_tiled_kernel(a_kcontig, bmode, use_stats, B, M, N, K, bias):
    """pa_tgemm_nn on LDS-resident weights (csrc/train_gemm_cm.hip: a workgroup pinned to a 128-row block of A, the B operand as 16-byte global loads
    that are the fragments of four interleaved column tiles, statistics in wave-private fp64 LDS blocks) against (a) the float64 statement of
    include/patchaugnet_hip.h's definition and (b) the LDS-tiled kernel on the same operands: every operand transform, both layouts of A,
    one- and two-half row blocks (M = 64 / 192 / 512), ragged tile counts per workgroup, bias, statistics."""
    from patchaugnet_amd import train_ops as T
    g = torch.Generator().manual_seed(M + 3 * N + K + bmode)
    A = (torch.randn(M, K, generator=g) if a_kcontig else torch.randn(K, M, generator=g)) / K ** 0.5
    X = torch.randn(B, K, N, generator=g)
    aux = torch.randn(B, K, N, generator=g)
    p = _p_block(K, g)
    bvec = torch.randn(M, generator=g)
    Ad, Xd, auxd, pd, bd = A.cuda(), X.cuda(), aux.cuda(), p.cuda().contiguous(), bvec.cuda()
    outs, sts = [], []
    try:
        for on in (0, 1):
            _cm_switch(on)
            C = torch.full((B, M, N), 0.5, device="cuda")
            st = torch.zeros(T.STAT_SLOTS, 2, M, dtype=torch.float64, device="cuda")
            T.tgemm_nn(B, M, N, K, Ad, 0, K if a_kcontig else M, a_kcontig, Xd, K * N, N, C, M * N, N, bmode=bmode, baux=auxd if bmode >= 2 else None,
                       bp=pd if bmode else None, bias=bd if bias else None, stats=st if use_stats else None)
            outs.append(C)
            sts.append(st.sum(0))
        torch.cuda.synchronize()
    finally:
        _cm_switch(-1)
    A64 = A.double() if a_kcontig else A.double().t()
    ref = torch.matmul(A64, torch.stack([_tf(bmode, X[b].double(), aux[b].double(), p.double()) for b in range(B)]))
    if bias:
        ref = ref + bvec.double()[None, :, None]
    scale = max(ref.abs().max().item(), 1.0)
    for name, C in zip(("lds-tiled", "lds-resident weights"), outs):
        err = (C.cpu().double() - ref).abs().max().item()
        assert err <= 2e-5 * scale + 1e-6 * K, (name, err, scale)
    assert (outs[0] - outs[1]).abs().max().item() <= 2e-5 * scale
    if use_stats:
        v = outs[1].cpu().double()
        s = torch.stack([v.sum((0, 2)), (v * v).sum((0, 2))])
        assert torch.allclose(sts[1].cpu(), s, rtol=1e-5, atol=1e-4 * scale), (sts[1].cpu() - s).abs().max()
        assert torch.allclose(sts[0], sts[1], rtol=1e-5, atol=1e-4 * scale * N)


@pytest.mark.parametrize("B,M,N,K,bmode", [(3, 64, 40960, 32, 1), (3, 64, 40960, 32, 0), (3, 64, 40960, 64, 1), (18, 64, 20480, 32, 1), (3, 128, 40960, 32, 1),
                                           (3, 256, 8192, 256, 1)])
def test_lds_resident_weights_gemm_short_tiles_repeated(B, M, N, K, bmode):
    """Tiles of 8 / 16 k-steps with 16-byte stores (K = 32 / 64, the first set-abstraction level at 8192-point clouds), twenty launches each
    compared with the LDS-tiled kernel element for element: a build without wait states behind its 16-byte buffer stores lost lanes 12..15 of a
    store's first dword about once in two thousand stores (gfx950 reads wide store data for two cycles after issue; the next row's gather wrote
    the same registers) -- invisible to a single launch of the other shapes."""
    from patchaugnet_amd import train_ops as T
    g = torch.Generator().manual_seed(K + M)
    A = (torch.randn(M, K, generator=g) / K ** 0.5).cuda()
    X = torch.randn(B, K, N, generator=g).cuda()
    p = (torch.rand(7, K, generator=g) + 0.25).cuda() if bmode else None
    try:
        _cm_switch(0)
        ref = torch.empty(B, M, N, device="cuda")
        T.tgemm_nn(B, M, N, K, A, 0, K, True, X, K * N, N, ref, M * N, N, bmode=bmode, bp=p)
        _cm_switch(1)
        scale = max(ref.abs().max().item(), 1.0)
        for rep in range(20):
            C = torch.zeros(B, M, N, device="cuda")
            T.tgemm_nn(B, M, N, K, A, 0, K, True, X, K * N, N, C, M * N, N, bmode=bmode, bp=p)
            bad = ((C - ref).abs() > 2e-5 * scale).sum().item()
            assert bad == 0, (rep, bad)
    finally:
        _cm_switch(-1)


@pytest.mark.parametrize("B,M,N,K,bmode,bias", [(18, 256, 4096, 64, 3, False), (4, 96, 4096, 32, 2, True), (3, 64, 8192, 128, 0, True), (2, 160, 2048, 256, 1, False)])
def test_lds_resident_weights_gemm_accumulating_into_c(B, M, N, K, bmode, bias):
    """beta = 1 (C += A . f(B) [+ bias]: the second gradient contribution of a tensor with two consumers, train_ops._NetVladFused) on the
    LDS-resident-weights kernel against the LDS-tiled one: whole and padded row blocks, with and without bias -- the guarded epilogue reads the
    four rows' old values and biases in one batch before its stores."""
    from patchaugnet_amd import train_ops as T
    g = torch.Generator().manual_seed(3 * M + K)
    A = (torch.randn(K, M, generator=g) / K ** 0.5).cuda()
    X, aux = torch.randn(B, K, N, generator=g).cuda(), torch.randn(B, K, N, generator=g).cuda()
    p = _p_block(K, g).cuda().contiguous()
    C0 = torch.randn(B, M, N, generator=g).cuda()
    bvec = torch.randn(M, generator=g).cuda()
    outs = []
    try:
        for on in (0, 1):
            _cm_switch(on)
            C = C0.clone()
            T.tgemm_nn(B, M, N, K, A, 0, M, False, X, K * N, N, C, M * N, N, bmode=bmode, baux=aux if bmode >= 2 else None, bp=p if bmode else None,
                       beta=1, bias=bvec if bias else None)
            outs.append(C)
        torch.cuda.synchronize()
    finally:
        _cm_switch(-1)
    scale = max(outs[0].abs().max().item(), 1.0)
    assert (outs[0] - outs[1]).abs().max().item() <= 2e-5 * scale
    assert (outs[0] - C0).abs().max().item() > 0.1                               # something was added


@pytest.mark.parametrize("B,M,N,K,relu_next", [(18, 32, 20480, 64, 1), (4, 96, 4096, 32, 1), (3, 256, 4096, 256, 0), (2, 160, 2048, 128, 1)])
def test_input_gradient_gemm_with_the_next_layers_bn_backward_sums_on_its_epilogue(B, M, N, K, relu_next):
    """pa_tgemm_nn_bnred (train_gemm.hip / train_gemm_cm.hip STATS 2: dX = W^T . bn_bwd(dY, Y) with the BatchNorm-backward sums of the layer that
    produced X accumulated on the epilogue) against the unfused pair it replaces (pa_tgemm_nn then pa_bn_bwd_reduce over (dX, raw X)), with the
    LDS-resident-weights kernel forced and with it off: row blocks with padding rows (M = 32, 96, 160: rows past M of the last cloud lie past the
    end of the tensors -- a build that read the next layer's raw output before the row guard faulted here), whole blocks, both ReLU settings."""
    from patchaugnet_amd import train_ops as T
    from patchaugnet_amd._lib import call, ptr
    g = torch.Generator().manual_seed(M + N + K)
    W = (torch.randn(K, M, generator=g) / K ** 0.5).cuda()                      # (O = K, C = M): A(m = c, k = o) = W[o * M + c]
    dY, Y = torch.randn(B, K, N, generator=g).cuda(), torch.randn(B, K, N, generator=g).cuda()
    p = _p_block(K, g).cuda().contiguous()
    ynext, pnext = torch.randn(B, M, N, generator=g).cuda(), _p_block(M, g).cuda().contiguous()
    res = []
    try:
        for on in (1, 0):
            _cm_switch(on)
            gp = torch.full((B, M, N), 7.0, device="cuda")
            sums = torch.zeros(2 * M, dtype=torch.float64, device="cuda")
            call("pa_tgemm_nn_bnred", B, M, N, K, ptr(W), M, 0, ptr(dY), K * N, N, 2, ptr(Y), ptr(p), ptr(gp), M * N, N, ptr(ynext), ptr(pnext), relu_next,
                 ptr(sums))
            res.append((gp, sums))
        _cm_switch(0)
        gp0 = torch.empty((B, M, N), device="cuda")
        T.tgemm_nn(B, M, N, K, W, 0, M, False, dY, K * N, N, gp0, M * N, N, bmode=2, baux=Y, bp=p)
        sums0 = torch.zeros(2 * M, dtype=torch.float64, device="cuda")
        call("pa_bn_bwd_reduce", B, M, N, ptr(gp0), ptr(ynext), ptr(pnext), relu_next, ptr(sums0), 0)
        torch.cuda.synchronize()
    finally:
        _cm_switch(-1)
    scale = max(gp0.abs().max().item(), 1.0)
    for gp, sums in res:
        assert (gp - gp0).abs().max().item() <= 2e-5 * scale
        assert torch.allclose(sums, sums0, rtol=1e-5, atol=1e-4 * scale * (B * N) ** 0.5), (sums - sums0).abs().max()


@pytest.mark.parametrize("B,C1,m,n,spec", [(2, 3, 256, 1024, [259, 256, 256]), (3, 3, 512, 2048, [67, 64, 32, 48]), (2, 8, 1024, 4096, [136, 128, 128]),
                                           (18, 3, 1024, 4096, [259, 256, 256, 256])])
def test_fp_level_with_the_first_layer_folded_through_the_interpolation(B, C1, m, n, spec):
    """backbone.FPModule under autograd with W [interp(F); S] = interp(W_a F) + W_b S (csrc/fp_fold_train.hip, train_ops._FoldedFPChain) against
    float64 torch autograd of the reference's statement (patch_aug_net.py:350-362: interpolation -> cat -> SharedMLP), in train() and eval()
    mode: output, dF, every parameter gradient (the first layer's interpolated AND skip columns), BatchNorm running statistics; and against the
    unfolded device path (FPModule.fold_first_layer = False) on the same inputs."""
    from patchaugnet_amd.backbone import FPModule
    torch.manual_seed(4)
    C2 = spec[0] - C1
    fp = FPModule(mlp=spec).cuda().train()
    with torch.no_grad():
        for l in fp.mlp:
            l.bn.bn.weight.uniform_(0.5, 1.5)
            l.bn.bn.bias.normal_(0, 0.2)
    unknown = torch.rand(B, n, 3, device="cuda") * 2 - 1
    known = unknown[:, torch.randperm(n)[:m]].contiguous()
    skip = torch.randn(B, C1, n, device="cuda")
    feats = torch.randn(B, C2, m, device="cuda")
    fp.train()
    idx, weight, lists = FPModule.geometry(unknown, known, lists_for_channels=C2)
    i64, w64 = idx.cpu().long(), weight.cpu().double()

    def run(mod, t):
        if t.is_cuda:
            return mod(unknown, known, skip, t, geo=(idx, weight, lists))
        interp = sum(w64[:, None, :, q] * t.gather(2, i64[:, None, :, q].expand(-1, t.shape[1], -1)) for q in range(3))
        return mod.mlp(torch.cat([interp, skip.cpu().double()], 1).unsqueeze(-1)).squeeze(-1)

    assert FPModule.fold_first_layer
    _compare(fp, feats, run)
    _randomize_running_stats(fp)
    _compare(fp, feats, run, train=False)
    # the unfolded device path on the same inputs (interpolation -> cat -> chain_train)
    fp.train()
    o1, dx1, g1, _ = _grads(fp, feats, run)
    try:
        FPModule.fold_first_layer = False
        o0, dx0, g0, _ = _grads(fp, feats, run)
    finally:
        FPModule.fold_first_layer = True
    # (device against device: an activation within rounding distance of zero may take the other ReLU branch in one of the two orders of summation,
    # which moves single gradient entries by O(1): the robust comparison of _compare, half of the elements within 20 tol and 1e-2 in relative L2)
    assert _rel(o1, o0) <= 2e-4
    ok, e = _rel_robust(dx1, dx0, 2e-4)
    assert ok, ("dF", e)
    for k in g0:
        ok, e = _rel_robust(g1[k], g0[k], 2e-4)
        assert ok, (k, e)
