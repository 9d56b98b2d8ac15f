"""Training-mode dense path on hand-written MFMA kernels (csrc/train_gemm.hip): forward AND backward.

What the reference runs as cuDNN/cuBLAS convolutions + BatchNorm + ReLU kernels and their autograd twins in ``model.train()``:

  * ``pt_util.SharedMLP``        utils/model_util/pt_util.py:16-41, :98-152   (1x1 conv -> BatchNorm(batch statistics) -> ReLU [-> max over k])
  * ``PointNetDecoder``          place_recognition/patch_aug_net/models/pointnet_autoencoder.py:85-111
  * ``NetVLADBase`` and the heads place_recognition/patch_aug_net/models/loupe.py:8-66, :159-222, :332-361

driven by ``train_place_recognition.py:142-169, :386-392``.  Activations stay CHANNEL-MAJOR ``(B, C, P)`` like the reference's, so a
1x1 convolution is, per cloud, ``Y (O x P) = W (O x C) . X (C x P)``.  BatchNorm needs the batch statistics between layers, so a layer
is one GEMM launch, but nothing elementwise is ever a pass of its own: the next GEMM's operand loader applies the previous layer's
BatchNorm + ReLU, the epilogue accumulates the statistics, and in the backward pass the loaders build the BatchNorm/ReLU input
gradient from (dZ, raw Y) on the fly.  Only the raw (pre-BatchNorm) layer outputs are kept for the backward pass.

Everything here calls libpatchaugnet_hip.so (pa_tgemm_nn / pa_tgemm_kk / pa_bn_*); there is no torch.matmul / rocBLAS / MIOpen in it.

The module path takes these functions for EVERY tensor on the MI355X -- train() or eval(), with or without autograd (eval(): BatchNorm
with the running statistics, ``training=False`` below) -- so a model on the device never reaches a library GEMM / convolution; the modules'
plain torch statements remain only as their CPU form (PointNetVLAD's CPU configuration, and the fp64 reference side of the tests).
"""
import torch
from torch.autograd import Function

from ._arena import zero_arena, zeros              # noqa: F401  (zero_arena is part of this module's interface)
from . import _lib as _lib_mod
from ._lib import call, check_device, ptr

KK_REPS = 32             # output replicas of pa_tgemm_kk_rep (small weight gradients over very long contractions)
STAT_SLOTS = 32          # PA_BN_STAT_SLOTS (include/patchaugnet_hip.h): replicas of a layer's statistics block


def on_device(x):
    """True when `x` lives on the MI355X: the dense layers of the module path then run on the HIP kernels of this file."""
    return x.is_cuda


def _guard(t):
    return torch.cuda.device(t.device)


def _f32(*ts):
    """The kernels are fp32: any other dtype on the device is refused (it would be read as raw fp32 words), never converted silently."""
    for t in ts:
        if t is not None and t.dtype != torch.float32:
            raise TypeError("patchaugnet_amd dense kernels are fp32 on the MI355X; got %s (run float64 references on CPU tensors)" % t.dtype)


# ---------------------------------------------------------------------------------------------------- thin wrappers of the C ABI
def tgemm_nn(batch, M, N, K, A, sAb, lda, a_kcontig, B, sBb, ldb, C, sCb, ldc, *, bmode=0, baux=None, bp=None, beta=0, bias=None, colv=None,
             act=0, stats=None, per_batch_stats=0):
    """C_b (M x N) = [beta C_b +] act(A_b (M x K) . f(B_b) (K x N) + bias[m]) -- include/patchaugnet_hip.h: pa_tgemm_nn."""
    call("pa_tgemm_nn", batch, M, N, K, ptr(A), sAb, lda, int(a_kcontig), ptr(B), sBb, ldb, bmode, ptr(baux), ptr(bp), ptr(C), sCb, ldc,
         int(beta), ptr(bias), ptr(colv), act, ptr(stats), int(per_batch_stats))


def tgemm_kk(batch, M, N, K, A, sAb, lda, B, sBb, ldb, C, sCb, ldc, *, amode=0, aaux=None, ap=None, bmode=0, bp=None, per_batch=0,
             per_batch_stats=0):
    """C (M x N) += sum_b sum_k fA(A_b)(m,k) fB(B_b)(n,k) -- include/patchaugnet_hip.h: pa_tgemm_kk."""
    if not per_batch and not per_batch_stats and M * N <= 512 and batch * K >= (1 << 18):
        # a small output over a very long contraction (first set-abstraction level: 32 x 6 over 18 x 20 480 points: 85 -> 54 us; from 32 x 32 outputs on the launch is bound by its operand traffic and the replicas gain nothing): partial tiles into
        # zero-filled replicas (out of the step's zero arena: no fill launch), then one small reduce -- instead of queueing atomics on M N addresses
        scratch = zeros((KK_REPS * M * N,), torch.float32, C.device)
        call("pa_tgemm_kk_rep", batch, M, N, K, ptr(A), sAb, lda, amode, ptr(aaux), ptr(ap), ptr(B), sBb, ldb, bmode, ptr(bp), ptr(C), ldc,
             ptr(scratch), KK_REPS)
        return
    call("pa_tgemm_kk", batch, M, N, K, ptr(A), sAb, lda, amode, ptr(aaux), ptr(ap), ptr(B), sBb, ldb, bmode, ptr(bp), ptr(C), sCb, ldc,
         int(per_batch), int(per_batch_stats))


class BNLayer:
    """One conv/linear + BatchNorm [+ ReLU] layer of a chain.  weight: (O, C[,1[,1]]) or, with transposed=True, (C, O)."""
    __slots__ = ("weight", "bias", "bn", "relu", "transposed")

    def __init__(self, weight, bn, bias=None, relu=True, transposed=False):
        self.weight, self.bias, self.bn, self.relu, self.transposed = weight, bias, bn, relu, transposed

    @property
    def out_channels(self):
        return self.weight.shape[1] if self.transposed else self.weight.shape[0]


def _bn_buffers(bn):
    if bn.track_running_stats and bn.running_mean is not None:
        # momentum None = cumulative moving average in torch; the hot path's layers all use the default 0.1
        assert bn.momentum is not None, "cumulative-average BatchNorm is not built"
        return bn.running_mean, bn.running_var, float(bn.momentum)
    return None, None, 0.0


class _ChainTrain(Function):
    """x (B, C0, P) -> L x [W . -> BatchNorm(batch stats) -> ReLU] [-> max over `pool` consecutive points].
    groups = True: every batch entry is its own BatchNorm batch (the decoder run over R related clouds in one set of launches)."""

    @staticmethod
    def forward(ctx, x, layers, pool, groups, training, *tensors):
        check_device(x)
        _f32(x, *tensors)
        B, cin, P = x.shape
        dev = x.device
        it = iter(tensors)
        Ws, biases, gammas, betas = [], [], [], []
        for L in layers:
            Ws.append(next(it))
            biases.append(next(it) if L.bias is not None else None)
            gammas.append(next(it))
            betas.append(next(it))
        G = B if groups else 1                    # statistics groups
        outs = [L.out_channels for L in layers]
        # one zero-filled arena for every layer's statistics replicas, one for the parameter blocks
        stats_all = zeros((G * STAT_SLOTS * 2 * sum(outs),), torch.float64, dev) if training else None
        p_all = torch.empty(G * 7 * sum(outs), dtype=torch.float32, device=dev)
        ys, ps = [], []
        prev, prevp = x, None
        so = po = 0
        with _guard(x):
            for i, L in enumerate(layers):
                W, O = Ws[i], outs[i]
                C = W.numel() // O
                assert C == cin, f"layer {i}: weight has {C} input channels, activation has {cin}"
                assert L.relu or i == len(layers) - 1, "only the last layer of a chain may come without ReLU"
                y = torch.empty((B, O, P), dtype=torch.float32, device=dev)
                stats = stats_all[so:so + G * STAT_SLOTS * 2 * O] if training else None
                p = p_all[po:po + G * 7 * O]
                so += G * STAT_SLOTS * 2 * O
                po += G * 7 * O
                tgemm_nn(B, O, P, C, W, 0, O if L.transposed else C, not L.transposed, prev, C * P, P, y, O * P, P,
                         bmode=0 if i == 0 else 1, bp=prevp, bias=biases[i], stats=stats, per_batch_stats=groups)
                if training:
                    rm, rv, mom = _bn_buffers(L.bn)
                    nbt = L.bn.num_batches_tracked if rm is not None else None           # += G inside the kernel
                    call("pa_bn_finalize", O, G, float(B * P // G), ptr(stats), ptr(gammas[i]), ptr(betas[i]), float(L.bn.eps), mom, ptr(rm), ptr(rv), ptr(p), ptr(nbt))
                else:       # eval(): the running statistics, nothing updated (torch.nn.BatchNorm in eval mode)
                    assert L.bn.running_mean is not None, "eval-mode BatchNorm without running statistics uses batch statistics: call with training=True"
                    call("pa_bn_eval_params", O, G, ptr(gammas[i]), ptr(betas[i]), ptr(L.bn.running_mean), ptr(L.bn.running_var), float(L.bn.eps), ptr(p))
                ys.append(y)
                ps.append(p)
                prev, prevp, cin = y, p, O
            Pout = P // pool if pool else P
            if pool:
                assert P % pool == 0
            out = torch.empty((B, cin, Pout), dtype=torch.float32, device=dev)
            arg = torch.empty((B, cin, Pout), dtype=torch.int8, device=dev) if pool else None
            call("pa_bn_apply", B, cin, P, int(pool), int(layers[-1].relu), ptr(prev), ptr(prevp), ptr(out), ptr(arg), int(groups))
        ctx.save_for_backward(x, *Ws)
        ctx.layers, ctx.pool, ctx.groups, ctx.ys, ctx.ps, ctx.arg = layers, pool, groups, ys, ps, arg
        ctx.bias_like, ctx.training = biases, training
        return out

    @staticmethod
    def backward(ctx, gout):
        x, *Ws = ctx.saved_tensors
        layers, pool, groups, ys, ps, arg = ctx.layers, ctx.pool, ctx.groups, ctx.ys, ctx.ps, ctx.arg
        B, _, P = x.shape
        dev = x.device
        G = B if groups else 1
        g = gout.contiguous()
        per_layer = [None] * len(layers)
        outs = [y.shape[1] for y in ys]
        ins = [x.shape[1]] + outs[:-1]
        sums_all = zeros((G * 2 * sum(outs),), torch.float64, dev)
        # weight gradients (split-K partial tiles are added with atomics: zero-filled) and dgamma / dbeta, one arena
        nW = [o * c for o, c in zip(outs, ins)]
        grads = zeros((sum(nW) + 2 * sum(outs),), torch.float32, dev, keep=True)        # parameter gradients: outlive the step
        wo = [0]
        for n in nW:
            wo.append(wo[-1] + n)
        go = wo[-1]
        so = [0]
        for o in outs:
            so.append(so[-1] + G * 2 * o)
        with _guard(x):
            reduced = False          # this layer's BatchNorm-backward sums already rode on the launch that produced its activation gradient
            if pool:
                O = outs[-1]
                full = torch.empty((B, O, P), dtype=torch.float32, device=dev)
                if groups:
                    call("pa_maxpool_bwd", B * O, P // pool, int(pool), ptr(g), ptr(arg), ptr(full))
                else:        # ... here: the pooled gradient is zero off the arg-max positions, so the last layer's sums need only those
                    call("pa_maxpool_bwd_bnred", B, O, P // pool, int(pool), ptr(g), ptr(arg), ptr(full), ptr(ys[-1]), ptr(ps[-1]), int(layers[-1].relu),
                         ptr(sums_all[so[-2]:so[-1]]))
                    reduced = True
                g = full
            for i in range(len(layers) - 1, -1, -1):
                L, W, y, p = layers[i], Ws[i], ys[i], ps[i]
                O, C = outs[i], ins[i]
                prev = x if i == 0 else ys[i - 1]
                sums = sums_all[so[i]:so[i + 1]]
                if not reduced:
                    call("pa_bn_bwd_reduce", B, O, P, ptr(g), ptr(y), ptr(p), int(L.relu), ptr(sums), int(groups))
                reduced = False
                dgamma = grads[go + 2 * sum(outs[:i]):go + 2 * sum(outs[:i]) + O]
                dbeta = grads[go + 2 * sum(outs[:i]) + O:go + 2 * sum(outs[:i]) + 2 * O]
                # eval(): no batch-statistics terms in the input gradient -- count = +inf leaves rows 4, 5 of p at zero
                call("pa_bn_bwd_finalize", O, G, float(B * P // G) if ctx.training else float("inf"), ptr(sums), ptr(p), ptr(dgamma), ptr(dbeta))
                mode = 2 if L.relu else 3
                dW = grads[wo[i]:wo[i + 1]].view(O, C)
                tgemm_kk(B, O, C, P, g, O * P, P, prev, C * P, P, dW, 0, C, amode=mode, aaux=y, ap=p,
                         bmode=0 if i == 0 else 1, bp=None if i == 0 else ps[i - 1], per_batch_stats=groups)
                if i > 0 or ctx.needs_input_grad[0]:
                    gp = torch.empty((B, C, P), dtype=torch.float32, device=dev)
                    # dX (C x P) = W^T (C x O) . dY (O x P): A(m = c, k = o) = W[o*C + c] (or W[c*O + o] for a transposed weight)
                    if i > 0 and not groups:
                        # ... together with layer i - 1's BatchNorm-backward sums over (gp, its raw output): one launch where the shape allows
                        call("pa_tgemm_nn_bnred", B, C, P, O, ptr(W), O if L.transposed else C, int(L.transposed), ptr(g), O * P, P, mode, ptr(y), ptr(p),
                             ptr(gp), C * P, P, ptr(ys[i - 1]), ptr(ps[i - 1]), int(layers[i - 1].relu), ptr(sums_all[so[i - 1]:so[i]]))
                        reduced = True
                    else:
                        acc = getattr(ctx, "accumulate_into", None) if i == 0 else None      # a second consumer's gradient of x already sits there
                        if acc is not None:
                            gp = acc
                        tgemm_nn(B, C, P, O, W, 0, O if L.transposed else C, L.transposed, g, O * P, P, gp, C * P, P, bmode=mode, baux=y, bp=p,
                                 beta=1 if acc is not None else 0, per_batch_stats=groups)
                    g = gp
                else:
                    g = None
                dWr = (dW.t().contiguous() if L.transposed else dW).view_as(W)
                # a bias in front of a training-mode BatchNorm has an identically zero gradient (the mean subtraction removes it); in
                # eval() it is d/dz of scale*z + shift summed over the points: scale * dbeta
                if ctx.bias_like[i] is None:
                    dbias = []
                elif ctx.training:
                    dbias = [torch.zeros_like(ctx.bias_like[i])]
                else:
                    dbias = [dbeta * p[:O]]
                per_layer[i] = [dWr] + dbias + [dgamma, dbeta]
        ctx.ys = ctx.ps = ctx.arg = None
        flat = [t for pl in per_layer for t in pl]
        return (g, None, None, None, None, *flat)


def chain_train(x, layers, pool=0, groups=False, training=True):
    """x: (B, C0, P) contiguous fp32 on the MI355X; layers: [BNLayer]; returns (B, C_L, P // pool or P).  training=False: BatchNorm with
    the running statistics (eval() mode), same kernels, forward and backward."""
    # every BatchNorm follows ITS OWN mode flag (a layer frozen with bn.eval() inside a model.train() model keeps its running statistics, as the
    # torch modules do); one launch chain runs one mode, so the layers of a chain have to agree
    modes = {bool(L.bn.training) for L in layers}
    if len(modes) > 1:
        raise NotImplementedError("chain_train: the BatchNorm layers of one chain are in different modes (train / eval); freeze the whole block")
    training = modes.pop() if modes else training
    tensors = []
    for L in layers:
        tensors.append(L.weight)
        if L.bias is not None:
            tensors.append(L.bias)
        tensors += [L.bn.weight, L.bn.bias]
    return _ChainTrain.apply(x.contiguous(), layers, int(pool), bool(groups), bool(training), *tensors)


class _FoldedFPChain(Function):
    """Feature-propagation level under autograd with its FIRST layer folded through the interpolation (csrc/fp_fold_train.hip):
    ``SharedMLP(cat([interpolation(F, idx, weight), S], 1))`` of patch_aug_net.py:350-362 without the (B, C2 + C1, n) tensor and with the first
    layer's three contractions on the m known points instead of the n >= 2 m unknown ones:

        W [interp(F); S] = interp(W_a F) + W_b S          (interpolation acts on the point axis, the 1x1 convolution on the channel axis)

    forward: Z = W_a F (pa_tgemm_nn on m columns) -> pa_fp_fold_forward (interpolation + skip term + the layer's BatchNorm statistics) -> the remaining
    layers as in _ChainTrain.  backward: the remaining layers as in _ChainTrain down to the gradient of the first layer's activation, then
    pa_bn_bwd_reduce / _finalize of that layer, pa_fp_fold_backward (G = interp^T(dY1), dW_b += dY1 S^T in one pass) and dW_a = G F^T,
    dF = W_a^T G on m columns.  S (the raw coordinates at the finest level) takes no gradient.  Same values as the unfolded form up to fp32
    summation order."""

    @staticmethod
    def forward(ctx, F, S, idx, weight, lists, layers, training, *tensors):
        check_device(F, S, idx, weight)
        _f32(F, S, *tensors)
        B, C2, m = F.shape
        C1, n = S.shape[1], S.shape[2]
        dev = F.device
        it = iter(tensors)
        Ws, gammas, betas = [], [], []
        for L in layers:
            assert L.bias is None and not L.transposed, "the folded feature-propagation chain takes SharedMLP layers (no bias, (O, C) weights)"
            Ws.append(next(it)); gammas.append(next(it)); betas.append(next(it))
        outs = [L.out_channels for L in layers]
        O = outs[0]
        W1 = Ws[0].reshape(O, C2 + C1)
        stats_all = zeros((STAT_SLOTS * 2 * sum(outs),), torch.float64, dev) if training else None
        p_all = torch.empty(7 * sum(outs), dtype=torch.float32, device=dev)
        ys, ps = [], []
        so = po = 0

        def bn_params(i, y_count):
            nonlocal so, po
            Oi = outs[i]
            stats = stats_all[so:so + STAT_SLOTS * 2 * Oi] if training else None
            p = p_all[po:po + 7 * Oi]
            so += STAT_SLOTS * 2 * Oi
            po += 7 * Oi
            return stats, p

        def finalize(i, stats, p):
            L = layers[i]
            if training:
                rm, rv, mom = _bn_buffers(L.bn)
                nbt = L.bn.num_batches_tracked if rm is not None else None
                call("pa_bn_finalize", outs[i], 1, float(B * n), ptr(stats), ptr(gammas[i]), ptr(betas[i]), float(L.bn.eps), mom, ptr(rm), ptr(rv), ptr(p), ptr(nbt))
            else:
                assert L.bn.running_mean is not None, "eval-mode BatchNorm without running statistics uses batch statistics: call with training=True"
                call("pa_bn_eval_params", outs[i], 1, ptr(gammas[i]), ptr(betas[i]), ptr(L.bn.running_mean), ptr(L.bn.running_var), float(L.bn.eps), ptr(p))

        with _guard(F):
            Wa = W1[:, :C2].contiguous()                                   # (O, C2): the interpolated channels' columns
            Z = torch.empty((B, O, m), dtype=torch.float32, device=dev)
            tgemm_nn(B, O, m, C2, Wa, 0, C2, True, F, C2 * m, m, Z, O * m, m)
            y = torch.empty((B, O, n), dtype=torch.float32, device=dev)
            stats, p = bn_params(0, n)
            Wb = W1[:, C2:]                                                # (O, C1) view, row stride C2 + C1: read in place
            call("pa_fp_fold_forward", B, O, m, n, C1, ptr(Z), ptr(idx), ptr(weight), ptr(S), ptr(Wb), C2 + C1, ptr(y), ptr(stats))
            finalize(0, stats, p)
            ys.append(y); ps.append(p)
            prev, prevp, cin = y, p, O
            for i in range(1, len(layers)):
                L, W, Oi = layers[i], Ws[i], outs[i]
                assert W.numel() // Oi == cin and (L.relu or i == len(layers) - 1)
                y = torch.empty((B, Oi, n), dtype=torch.float32, device=dev)
                stats, p = bn_params(i, n)
                tgemm_nn(B, Oi, n, cin, W, 0, cin, True, prev, cin * n, n, y, Oi * n, n, bmode=1, bp=prevp, stats=stats)
                finalize(i, stats, p)
                ys.append(y); ps.append(p)
                prev, prevp, cin = y, p, Oi
            out = torch.empty((B, cin, n), dtype=torch.float32, device=dev)
            call("pa_bn_apply", B, cin, n, 0, int(layers[-1].relu), ptr(prev), ptr(prevp), ptr(out), ptr(None), 0)
        ctx.save_for_backward(F, S, idx, weight, Wa, *Ws)
        ctx.layers, ctx.ys, ctx.ps, ctx.lists, ctx.training = layers, ys, ps, lists, training
        return out

    @staticmethod
    def backward(ctx, gout):
        F, S, idx, weight, Wa, *Ws = ctx.saved_tensors
        layers, ys, ps, lists = ctx.layers, ctx.ys, ctx.ps, ctx.lists
        B, C2, m = F.shape
        C1, n = S.shape[1], S.shape[2]
        dev = F.device
        outs = [y.shape[1] for y in ys]
        ins = [C2 + C1] + outs[:-1]
        O = outs[0]
        g = gout.contiguous()
        sums_all = zeros((2 * sum(outs),), torch.float64, dev)
        nW = [o * c for o, c in zip(outs, ins)]
        grads = zeros((sum(nW) + 2 * sum(outs),), torch.float32, dev, keep=True)        # parameter gradients: outlive the step
        wo = [0]
        for k in nW:
            wo.append(wo[-1] + k)
        go = wo[-1]
        so = [0]
        for o in outs:
            so.append(so[-1] + 2 * o)
        per_layer = [None] * len(layers)
        count = float(B * n) if ctx.training else float("inf")
        with _guard(F):
            reduced = False
            for i in range(len(layers) - 1, -1, -1):
                L, y, p = layers[i], ys[i], ps[i]
                Oi, Ci = outs[i], ins[i]
                sums = sums_all[so[i]:so[i + 1]]
                if not reduced:
                    call("pa_bn_bwd_reduce", B, Oi, n, ptr(g), ptr(y), ptr(p), int(L.relu), ptr(sums), 0)
                dgamma = grads[go + 2 * sum(outs[:i]):go + 2 * sum(outs[:i]) + Oi]
                dbeta = grads[go + 2 * sum(outs[:i]) + Oi:go + 2 * sum(outs[:i]) + 2 * Oi]
                call("pa_bn_bwd_finalize", Oi, 1, count, ptr(sums), ptr(p), ptr(dgamma), ptr(dbeta))
                mode = 2 if L.relu else 3
                dW = grads[wo[i]:wo[i + 1]].view(Oi, Ci)
                if i > 0:
                    tgemm_kk(B, Oi, Ci, n, g, Oi * n, n, ys[i - 1], Ci * n, n, dW, 0, Ci, amode=mode, aaux=y, ap=p, bmode=1, bp=ps[i - 1])
                    gp = torch.empty((B, Ci, n), dtype=torch.float32, device=dev)
                    call("pa_tgemm_nn_bnred", B, Ci, n, Oi, ptr(Ws[i]), Ci, 0, ptr(g), Oi * n, n, mode, ptr(y), ptr(p), ptr(gp), Ci * n, n,
                         ptr(ys[i - 1]), ptr(ps[i - 1]), int(layers[i - 1].relu), ptr(sums_all[so[i - 1]:so[i]]))
                    reduced = True
                    g = gp
                else:
                    if lists is None:
                        lists = torch.empty(_lib_mod.lib().pa_interpolation_backward_scratch_ints(B, n, m), dtype=torch.int32, device=dev)
                        call("pa_interpolation_backward_lists", B, n, m, ptr(idx), ptr(weight), ptr(lists))
                    G = torch.empty((B, O, m), dtype=torch.float32, device=dev)
                    # dW[:, C2:] += dY1 S^T (the kernel's atomics, row stride C2 + C1); G = interp^T(dY1)
                    call("pa_fp_fold_backward", B, O, n, m, C1, ptr(g), ptr(y), ptr(p), int(L.relu), ptr(S), ptr(lists), ptr(G), ptr(dW[:, C2:]), C2 + C1)
                    # dW[:, :C2] += sum_b G_b F_b^T (contraction over the m known points); dF_b = W_a^T G_b
                    tgemm_kk(B, O, C2, m, G, O * m, m, F, C2 * m, m, dW, 0, C2 + C1)
                    dF = None
                    if ctx.needs_input_grad[0]:
                        dF = torch.empty((B, C2, m), dtype=torch.float32, device=dev)
                        tgemm_nn(B, C2, m, O, Wa, 0, C2, False, G, O * m, m, dF, C2 * m, m)
                per_layer[i] = [dW.view_as(Ws[i]), dgamma, dbeta]
        ctx.ys = ctx.ps = None
        flat = [t for pl in per_layer for t in pl]
        return (dF, None, None, None, None, None, None, *flat)


def fp_chain_train_folded(known_feats, skip, idx, weight, lists, layers, training=True):
    """The decoder level ``SharedMLP(cat([interpolation(known_feats, idx, weight), skip], 1))`` with its first layer folded through the
    interpolation (see _FoldedFPChain).  known_feats (B, C2, m), skip (B, C1 <= 8, n) without gradient, idx / weight (B, n, 3)."""
    modes = {bool(L.bn.training) for L in layers}
    if len(modes) > 1:
        raise NotImplementedError("fp_chain_train_folded: the BatchNorm layers of one chain are in different modes (train / eval); freeze the whole block")
    training = modes.pop() if modes else training
    tensors = []
    for L in layers:
        tensors += [L.weight, L.bn.weight, L.bn.bias]
    return _FoldedFPChain.apply(known_feats.contiguous(), skip.contiguous(), idx, weight, lists, layers, bool(training), *tensors)


def fp_fold_applies(known_feats, skip, idx, n_layers):
    """Shape rule of the fold (a function of the level's architecture, never of the batch): a skip of at most eight channels that takes no
    gradient (the raw coordinates of the finest level), at least twice as many unknown as known points, shapes the kernels are built for."""
    if skip is None or known_feats.dim() != 3 or skip.dim() != 3 or not known_feats.is_cuda:
        return False
    B, C2, m = known_feats.shape
    C1, n = skip.shape[1], skip.shape[2]
    return (n_layers >= 2 and 1 <= C1 <= 8 and not skip.requires_grad and n >= 2 * m and 1024 <= n <= 4096 and n % 4 == 0 and m % 4 == 0 and m <= 4096
            and C2 % 4 == 0 and known_feats.dtype == torch.float32 and skip.dtype == torch.float32)


class _LinearCM(Function):
    """Y_b (O x P) = act(W (O x C) . X_b (C x P) + bias); act 0 none / 1 tanh.  W (O, C[,1]) shared by the batch."""

    @staticmethod
    def forward(ctx, x, W, bias, act):
        check_device(x, W)
        _f32(x, W, bias)
        B, C, P = x.shape
        O = W.shape[0]
        assert W.numel() == O * C
        y = torch.empty((B, O, P), dtype=torch.float32, device=x.device)
        with _guard(x):
            tgemm_nn(B, O, P, C, W, 0, C, True, x, C * P, P, y, O * P, P, bias=bias, act=act)
        ctx.save_for_backward(x, W, y if act else None)
        ctx.act, ctx.has_bias = act, bias is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        x, W, y = ctx.saved_tensors
        B, C, P = x.shape
        O = W.shape[0]
        g = gy.contiguous()
        if ctx.act == 1:
            g = g * (1.0 - y * y)
        dx = dW = db = None
        with _guard(x):
            if ctx.needs_input_grad[0]:
                dx = torch.empty_like(x)
                tgemm_nn(B, C, P, O, W, 0, C, False, g, O * P, P, dx, C * P, P)
            if ctx.needs_input_grad[1]:
                dW = zeros((O, C), torch.float32, x.device, keep=True)
                tgemm_kk(B, O, C, P, g, O * P, P, x, C * P, P, dW, 0, C)
                dW = dW.view_as(W)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = g.sum(dim=(0, 2))
        return dx, dW, db, None


def linear_cm(x, W, bias=None, act=0):
    return _LinearCM.apply(x.contiguous(), W, bias, act)


class _BmmNT(Function):
    """C_b (M x N) = A_b (M x K) . B_b (N x K)^T, both operands contiguous along the contraction (NetVLAD: X (C x n) . act (K x n)^T)."""

    @staticmethod
    def forward(ctx, a, b):
        check_device(a, b)
        _f32(a, b)
        batch, M, K = a.shape
        N = b.shape[1]
        assert b.shape[0] == batch and b.shape[2] == K
        c = zeros((batch, M, N), torch.float32, a.device)
        with _guard(a):
            tgemm_kk(batch, M, N, K, a, M * K, K, b, N * K, K, c, M * N, N, per_batch=1)
        ctx.save_for_backward(a, b)
        return c

    @staticmethod
    def backward(ctx, gc):
        a, b = ctx.saved_tensors
        batch, M, K = a.shape
        N = b.shape[1]
        g = gc.contiguous()
        da = db = None
        with _guard(a):
            if ctx.needs_input_grad[0]:      # dA (M x K) = dC (M x N) . B (N x K)
                da = torch.empty_like(a)
                tgemm_nn(batch, M, K, N, g, M * N, N, True, b, N * K, K, da, M * K, K)
            if ctx.needs_input_grad[1]:      # dB (N x K) = dC^T (N x M) . A (M x K)
                db = torch.empty_like(b)
                tgemm_nn(batch, N, K, M, g, M * N, N, False, a, M * K, K, db, N * K, K)
        return da, db


def bmm_nt(a, b):
    return _BmmNT.apply(a.contiguous(), b.contiguous())


class _BmmNN(Function):
    """C_b (M x N) = A_b (M x K) . B_b (K x N), row-major operands (grouped self-attention: x_v (C x n) . attn (n x n))."""

    @staticmethod
    def forward(ctx, a, b):
        check_device(a, b)
        _f32(a, b)
        batch, M, K = a.shape
        N = b.shape[2]
        assert b.shape[0] == batch and b.shape[1] == K
        c = torch.empty((batch, M, N), dtype=torch.float32, device=a.device)
        with _guard(a):
            tgemm_nn(batch, M, N, K, a, M * K, K, True, b, K * N, N, c, M * N, N)
        ctx.save_for_backward(a, b)
        return c

    @staticmethod
    def backward(ctx, gc):
        a, b = ctx.saved_tensors
        batch, M, K = a.shape
        N = b.shape[2]
        g = gc.contiguous()
        da = db = None
        with _guard(a):
            if ctx.needs_input_grad[0]:      # dA (M x K) = dC (M x N) . B (K x N)^T: both contiguous along n
                da = zeros(a.shape, torch.float32, a.device)
                tgemm_kk(batch, M, K, N, g, M * N, N, b, K * N, N, da, M * K, K, per_batch=1)
            if ctx.needs_input_grad[1]:      # dB (K x N) = A^T (K x M) . dC (M x N)
                db = torch.empty_like(b)
                tgemm_nn(batch, K, N, M, a, M * K, K, False, g, M * N, N, db, K * N, N)
        return da, db


def bmm_nn(a, b):
    return _BmmNN.apply(a.contiguous(), b.contiguous())


class _GramTN(Function):
    """E_b (N x N) = Y_b^T Y_b for Y (B, C, N): the attention energy (the sum over the groups of the per-group Grams, pptnet.py:273-275)."""

    @staticmethod
    def forward(ctx, y):
        check_device(y)
        _f32(y)
        batch, C, N = y.shape
        e = torch.empty((batch, N, N), dtype=torch.float32, device=y.device)
        with _guard(y):
            tgemm_nn(batch, N, N, C, y, C * N, N, False, y, C * N, N, e, N * N, N)      # A(m, k) = Y[k][m]
        ctx.save_for_backward(y)
        return e

    @staticmethod
    def backward(ctx, ge):
        (y,) = ctx.saved_tensors
        batch, C, N = y.shape
        g = ge.contiguous()
        dy = zeros(y.shape, torch.float32, y.device)
        with _guard(y):
            # dY = Y (dE + dE^T):  Y . dE^T (contiguous along the contraction: split-K kernel, accumulates into the zero-filled dy) ...
            tgemm_kk(batch, C, N, N, y, C * N, N, g, N * N, N, dy, C * N, N, per_batch=1)
            # ... + Y . dE (beta = 1)
            tgemm_nn(batch, C, N, N, y, C * N, N, True, g, N * N, N, dy, C * N, N, beta=1)
        return dy


class _SoftmaxRenorm(Function):
    """A = softmax_rows(E) / (1e-9 + column sums) (pptnet.py:276-277) on the kernels of csrc/attention_train.hip; E is consumed (in place)."""

    @staticmethod
    def forward(ctx, e):
        check_device(e)
        _f32(e)
        from . import _lib
        batch, N, _ = e.shape
        a = e                                                    # in place: the energy is dead after the soft-max
        colsum = torch.empty((batch, N), dtype=torch.float32, device=e.device)
        scratch = torch.empty(_lib.lib().pa_attn_train_scratch_floats(batch, N), dtype=torch.float32, device=e.device)
        with _guard(e):
            call("pa_attn_softmax_renorm", batch, N, ptr(a), ptr(colsum), ptr(scratch))
        ctx.mark_dirty(e)
        ctx.save_for_backward(a, colsum)
        return a

    @staticmethod
    def backward(ctx, ga):
        from . import _lib
        a, colsum = ctx.saved_tensors
        batch, N, _ = a.shape
        g = ga.contiguous().clone()                              # becomes dE in place
        scratch = torch.empty(_lib.lib().pa_attn_train_scratch_floats(batch, N) + batch * N, dtype=torch.float32, device=a.device)
        with _guard(a):
            call("pa_attn_softmax_renorm_backward", batch, N, ptr(a), ptr(colsum), ptr(g), ptr(scratch))
        return g


def sa_attention_train(y, x_v):
    """Grouped self-attention core under autograd (pptnet.py:273-278): y = k_conv(x) (B, C, N) (q and k are tied), x_v = v_conv(x) (B, C, N)
    -> x_r = x_v @ A, A = column-renormalised row soft-max of Y^T Y.  GEMMs on train_gemm.hip, the rest on attention_train.hip; one
    (B, N, N) matrix is kept for the backward pass."""
    attn = _SoftmaxRenorm.apply(_GramTN.apply(y.contiguous()))
    return bmm_nn(x_v, attn)


class _LinearRows(Function):
    """Y (R x O) = X (R x K) . W (O x K)^T + bias -- nn.Linear on a few rows and a long contraction (APFA's 21504 -> 256 FC): split-K."""

    @staticmethod
    def forward(ctx, x, W, bias):
        check_device(x, W)
        _f32(x, W, bias)
        R, K = x.shape
        O = W.shape[0]
        # the split-K kernel ACCUMULATES into y: start from the bias in a buffer of its own (expand(1, O).contiguous() would alias the parameter)
        y = bias.detach().expand(R, O).clone() if bias is not None else torch.zeros((R, O), dtype=torch.float32, device=x.device)
        with _guard(x):
            tgemm_kk(1, R, O, K, x, 0, K, W, 0, K, y, 0, O)
        ctx.save_for_backward(x, W)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        x, W = ctx.saved_tensors
        R, K = x.shape
        O = W.shape[0]
        g = gy.contiguous()
        dx = dW = db = None
        with _guard(x):
            if ctx.needs_input_grad[0]:      # dX (R x K) = dY (R x O) . W (O x K)
                dx = torch.empty_like(x)
                tgemm_nn(1, R, K, O, g, 0, O, True, W, 0, K, dx, 0, K)
            if ctx.needs_input_grad[1]:      # dW (O x K) = dY^T (O x R) . X (R x K)
                dW = torch.empty_like(W)
                tgemm_nn(1, O, K, R, g, 0, O, False, x, 0, K, dW, 0, K)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = g.sum(0)
        return dx, dW, db


def linear_rows(x, W, bias=None):
    return _LinearRows.apply(x.contiguous(), W, bias)


class _MatmulRows(Function):
    """Y (R x N) = X (R x K) . W (K x N)  (context gating, the FC heads of aggregation type 0 / PPT-Net)."""

    @staticmethod
    def forward(ctx, x, W):
        check_device(x, W)
        _f32(x, W)
        R, K = x.shape
        N = W.shape[1]
        y = torch.empty((R, N), dtype=torch.float32, device=x.device)
        with _guard(x):
            tgemm_nn(1, R, N, K, x, 0, K, True, W, 0, N, y, 0, N)
        ctx.save_for_backward(x, W)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, W = ctx.saved_tensors
        R, K = x.shape
        N = W.shape[1]
        g = gy.contiguous()
        dx = dW = None
        with _guard(x):
            if ctx.needs_input_grad[0]:      # dX (R x K) = dY (R x N) . W (K x N)^T
                dx = zeros(x.shape, torch.float32, x.device)
                tgemm_kk(1, R, K, N, g, 0, N, W, 0, N, dx, 0, K)
            if ctx.needs_input_grad[1]:      # dW (K x N) = X^T (K x R) . dY (R x N)
                dW = torch.empty_like(W)
                tgemm_nn(1, K, N, R, x, 0, K, False, g, 0, N, dW, 0, N)
        return dx, dW


def matmul_rows(x, W):
    return _MatmulRows.apply(x.contiguous(), W)


class _NetVladTail(Function):
    """NetVLAD after the assignment GEMM + BatchNorm (loupe.py:207-221): pre (B, K, N) logits, x (B, C, N) features, cw2 (1, C, K) ->
    normalise_C(x . act^T - a_sum * cw2), act = softmax_K(pre), a_sum = act summed over the points.  Two launches of csrc/train_glue.hip around the
    MFMA GEMM forward, three around the two GEMMs backward (the torch statement: 7 launches forward, ~20 backward)."""

    @staticmethod
    def forward(ctx, pre, x, cw2):
        check_device(pre, x, cw2)
        _f32(pre, x, cw2)
        B, K, N = pre.shape
        C = x.shape[1]
        assert x.shape == (B, C, N) and cw2.numel() == C * K
        dev = pre.device
        nblk = (N + 255) // 256
        act = torch.empty_like(pre)
        part = torch.empty((B, nblk, K), dtype=torch.float32, device=dev)
        raw = zeros((B, C, K), torch.float32, dev)                               # split-K accumulates
        out = torch.empty((B, C, K), dtype=torch.float32, device=dev)
        asum = torch.empty((B, K), dtype=torch.float32, device=dev)
        nrm = torch.empty((B, K), dtype=torch.float32, device=dev)
        with _guard(pre):
            call("pa_softmax_cols", B, K, N, ptr(pre), ptr(act), ptr(part))
            tgemm_kk(B, C, K, N, x, C * N, N, act, K * N, N, raw, C * K, K, per_batch=1)
            call("pa_vlad_residual_normalize", B, C, K, nblk, ptr(raw), ptr(part), ptr(cw2), ptr(out), ptr(asum), ptr(nrm))
        ctx.save_for_backward(x, act, cw2, out, asum, nrm)
        return out

    @staticmethod
    def backward(ctx, gout):
        x, act, cw2, out, asum, nrm = ctx.saved_tensors
        B, K, N = act.shape
        C = x.shape[1]
        g = gout.contiguous()
        dv = torch.empty_like(out)
        dasum = torch.empty_like(asum)
        dcw2 = torch.empty_like(cw2) if ctx.needs_input_grad[2] else None
        dx = dpre = None
        with _guard(x):
            call("pa_vlad_residual_normalize_backward", B, C, K, ptr(g), ptr(out), ptr(nrm), ptr(asum), ptr(cw2), ptr(dv), ptr(dasum), ptr(dcw2))
            if ctx.needs_input_grad[1]:      # dX (C x N) = dV (C x K) . act (K x N)
                dx = torch.empty_like(x)
                tgemm_nn(B, C, N, K, dv, C * K, K, True, act, K * N, N, dx, C * N, N)
            if ctx.needs_input_grad[0]:      # dact (K x N) = dV^T (K x C) . X (C x N), then through the soft-max in place
                dpre = torch.empty_like(act)
                tgemm_nn(B, K, N, C, dv, C * K, K, False, x, C * N, N, dpre, K * N, N)
                call("pa_softmax_cols_backward", B, K, N, ptr(act), ptr(dpre), ptr(dasum), ptr(dpre))
        return dpre, dx, dcw2


class _Bag:
    """Stand-in for an autograd ctx when one Function drives another Function's static forward / backward."""

    def save_for_backward(self, *tensors):
        self.saved_tensors = tensors


class _NetVladFused(Function):
    """NetVLADBase.forward (loupe.py:196-222) as ONE autograd node: the assignment layer (_ChainTrain, one BatchNorm layer without ReLU) and the
    tail (_NetVladTail) both consume x; as two nodes autograd adds their two (B, C, N) gradients with a pass of its own (75 MB read twice and
    written once at the finest scale).  Here the tail's backward writes dx and the assignment layer's input-gradient contraction ADDS to it
    (pa_tgemm_nn beta = 1)."""

    @staticmethod
    def forward(ctx, x, layer, training, cw2, W, gamma, beta):
        c1, c2 = _Bag(), _Bag()
        pre = _ChainTrain.forward(c1, x, [layer], 0, False, training, W, gamma, beta)
        out = _NetVladTail.forward(c2, pre, x, cw2)
        # the two bags' tensors go through autograd's own saved-tensor mechanism (version check on an in-place change between forward and
        # backward, saved-tensor hooks, release with the graph); the bags keep only their non-tensor attributes
        t1, t2 = tuple(getattr(c1, "saved_tensors", ())), tuple(getattr(c2, "saved_tensors", ()))
        ctx.save_for_backward(*t1, *t2)
        ctx.n1 = len(t1)
        c1.saved_tensors = c2.saved_tensors = ()
        ctx.c1, ctx.c2 = c1, c2
        return out

    @staticmethod
    def backward(ctx, gout):
        c1, c2 = ctx.c1, ctx.c2
        saved = ctx.saved_tensors
        c1.saved_tensors, c2.saved_tensors = saved[:ctx.n1], saved[ctx.n1:]
        need_x = ctx.needs_input_grad[0]
        c2.needs_input_grad = (True, need_x, ctx.needs_input_grad[3])
        dpre, dx, dcw2 = _NetVladTail.backward(c2, gout)
        c1.needs_input_grad = (need_x,)
        c1.accumulate_into = dx
        res = _ChainTrain.backward(c1, dpre)
        ctx.c1 = ctx.c2 = None
        return res[0], None, None, dcw2, res[5], res[6], res[7]


def netvlad_fused(x, layer, cw2, training=True):
    """x (B, C, N) -> intra-normalised VLAD (B, C, K): assignment layer + tail as one autograd node (see _NetVladFused)."""
    training = bool(layer.bn.training)
    return _NetVladFused.apply(x.contiguous(), layer, training, cw2, layer.weight, layer.bn.weight, layer.bn.bias)


def netvlad_tail(pre, x, cw2):
    return _NetVladTail.apply(pre.contiguous(), x.contiguous(), cw2.contiguous())


class _L2Normalize(Function):
    """torch.nn.functional.normalize(x, dim=1) of a (B, C) or (B, C, M) tensor, one launch each way (csrc/train_glue.hip)."""

    @staticmethod
    def forward(ctx, x):
        check_device(x)
        _f32(x)
        B, C = x.shape[:2]
        M = x.numel() // (B * C)
        out = torch.empty_like(x)
        nrm = torch.empty((B, M), dtype=torch.float32, device=x.device)
        with _guard(x):
            call("pa_l2_normalize", B, C, M, ptr(x), ptr(out), ptr(nrm))
        ctx.save_for_backward(out, nrm)
        return out

    @staticmethod
    def backward(ctx, gout):
        out, nrm = ctx.saved_tensors
        B, C = out.shape[:2]
        g = gout.contiguous()
        dx = torch.empty_like(out)
        with _guard(out):
            call("pa_l2_normalize_backward", B, C, nrm.shape[1], ptr(g), ptr(out), ptr(nrm), ptr(dx))
        return dx


def l2_normalize(x):
    """F.normalize(x, dim=1) (p = 2, eps = 1e-12) for x (B, C) or (B, C, ...)."""
    return _L2Normalize.apply(x.contiguous())


class _AfaAttention(Function):
    """MLPAttentionLayer's tail (loupe.py:27-41): r = conv(x) (B, C, K) -> w = softmax_K(max_C r), out = relu(x + x * w)."""

    @staticmethod
    def forward(ctx, x, r):
        check_device(x, r)
        _f32(x, r)
        B, C, K = x.shape
        out = torch.empty_like(x)
        w = torch.empty((B, K), dtype=torch.float32, device=x.device)
        arg = torch.empty((B, K), dtype=torch.int32, device=x.device)
        with _guard(x):
            call("pa_afa_attention", B, C, K, ptr(x), ptr(r), ptr(out), ptr(w), ptr(arg))
        ctx.save_for_backward(x, w, arg)
        return out

    @staticmethod
    def backward(ctx, gout):
        x, w, arg = ctx.saved_tensors
        B, C, K = x.shape
        g = gout.contiguous()
        dx = torch.empty_like(x)
        dr = torch.empty_like(x) if ctx.needs_input_grad[1] else None
        with _guard(x):
            call("pa_afa_attention_backward", B, C, K, ptr(g), ptr(x), ptr(w), ptr(arg), ptr(dx), ptr(dr))
        return dx, dr


def afa_attention(x, r):
    return _AfaAttention.apply(x.contiguous(), r.contiguous())


class _BNRowsTrain(Function):
    """torch.nn.BatchNorm1d in train mode over the rows of a small (R, F) matrix (the 256-wide heads: R = clouds in the step): one launch each way,
    running statistics and num_batches_tracked updated by the forward kernel like the torch module does."""

    @staticmethod
    def forward(ctx, x, gamma, beta, bn):
        check_device(x)
        _f32(x, gamma, beta)
        R, F_ = x.shape
        out = torch.empty_like(x)
        mean = torch.empty(F_, dtype=torch.float32, device=x.device)
        rstd = torch.empty(F_, dtype=torch.float32, device=x.device)
        rm, rv, mom = _bn_buffers(bn)
        nbt = bn.num_batches_tracked if (rm is not None and bn.num_batches_tracked is not None) else None
        with _guard(x):
            call("pa_bn_rows_train", R, F_, ptr(x), ptr(gamma), ptr(beta), float(bn.eps), mom, ptr(rm), ptr(rv), ptr(nbt), ptr(out), ptr(mean), ptr(rstd))
        ctx.save_for_backward(x, mean, rstd, gamma)
        return out

    @staticmethod
    def backward(ctx, gout):
        x, mean, rstd, gamma = ctx.saved_tensors
        R, F_ = x.shape
        g = gout.contiguous()
        dx = torch.empty_like(x)
        dgamma = torch.empty(F_, dtype=torch.float32, device=x.device) if gamma is not None and ctx.needs_input_grad[1] else None
        dbeta = torch.empty(F_, dtype=torch.float32, device=x.device) if ctx.needs_input_grad[2] else None
        with _guard(x):
            call("pa_bn_rows_backward", R, F_, ptr(g), ptr(x), ptr(mean), ptr(rstd), ptr(gamma), ptr(dx), ptr(dgamma), ptr(dbeta))
        return dx, dgamma, dbeta, None


def bn_rows(bn, x, training):
    """BatchNorm1d over the rows of a small (R, F) matrix in the module's mode (elementwise: no dense kernel either way)."""
    if training or bn.running_mean is None:          # track_running_stats=False: batch statistics in either mode, like torch
        if x.shape[0] == 1:
            raise ValueError("Expected more than 1 value per channel when training, got input size %s" % (tuple(x.shape),))
        return bn_rows_train(bn, x)
    y = (x - bn.running_mean) * torch.rsqrt(bn.running_var + bn.eps)
    if bn.weight is not None:
        y = y * bn.weight
    if bn.bias is not None:
        y = y + bn.bias
    return y


def bn_rows_train(bn, x):
    """BatchNorm1d in train mode over the rows of a small (R, F) matrix (csrc/train_glue.hip)."""
    return _BNRowsTrain.apply(x.contiguous(), bn.weight, bn.bias, bn)
