"""Fused inference engine for PatchAugNet on MI355X (evaluation mode).

Same parameters as the nn.Module tree (patch_aug_net.Network), different execution plan:

  * activations are point-major (B*N, C) so neighbour gathers read whole contiguous rows;
  * BatchNorm (eval) is folded into the 1x1-conv weights once, weights are stored K-major for the MFMA B operand;
  * each set-abstraction level is ONE kernel (gather + centre-subtract + concat + 3-layer shared MLP + max over the
    neighbourhood, csrc/mlp_chain.hip) and each feature-propagation level is ONE kernel (3-NN interpolation + concat
    + shared MLP); the (B, C, m, k) grouped tensors of the reference (pointops.py:559-570, patch_aug_net.py:234-237)
    are never materialised;
  * the kNN dilation of the reference keeps the nsample NEAREST of dilation*nsample candidates in a random order
    (pointops.py:553-555); the max over the neighbourhood is order-invariant, so the engine
    asks for nsample neighbours directly and skips the permutation.

What the reference computes per op is cited in csrc/*.hip; parity is checked in tests/test_gpu_models.py against the
reference-generated golden vectors and the CPU oracle.
"""
import ctypes

import torch
import torch.nn.functional as F

from . import _lib
from ._lib import call, ptr

_F32P = ctypes.POINTER(ctypes.c_float)


def fold_shared_mlp(mlp, device):
    """[(Wt (kpad, n) K-major, bias (n,), k, kpad, n)] for each Conv1x1+BN+ReLU layer, BatchNorm folded (fp64 math)."""
    out = []
    for layer in mlp.children():
        w = layer.conv.weight.detach().double().flatten(1)            # (n, k)
        bn = layer.bn.bn
        scale = bn.weight.detach().double() / torch.sqrt(bn.running_var.detach().double() + bn.eps)
        shift = bn.bias.detach().double() - bn.running_mean.detach().double() * scale
        n, k = w.shape
        if n % 16:
            raise ValueError(f"fused engine needs output widths that are multiples of 16, got {n}")
        kpad = (k + 3) // 4 * 4
        wt = torch.zeros(kpad, n, dtype=torch.float64)
        wt[:k] = (w * scale[:, None]).t()
        out.append((wt.float().contiguous().to(device), shift.float().contiguous().to(device), k, kpad, n))
    return out


class _Chain:
    """Host-side descriptor of one pa_mlp_chain call (pointer arrays are built once)."""

    def __init__(self, layers):
        self.layers = layers
        n = len(layers)
        self.n = n
        self.wt = (ctypes.c_void_p * n)(*[l[0].data_ptr() for l in layers])
        self.bias = (ctypes.c_void_p * n)(*[l[1].data_ptr() for l in layers])
        self.kpad = (ctypes.c_int * n)(*[l[3] for l in layers])
        self.nout = (ctypes.c_int * n)(*[l[4] for l in layers])
        self.k0 = layers[0][2]
        self.n_last = layers[-1][4]
        self.hidden_ok_pooled = all(l[4] <= 64 for l in layers[:-1])

    def _common(self):
        return (self.n, ctypes.cast(self.wt, ctypes.c_void_p), ctypes.cast(self.bias, ctypes.c_void_p),
                ctypes.cast(self.kpad, ctypes.c_void_p), ctypes.cast(self.nout, ctypes.c_void_p))

    def sa(self, xyz, feat, center_idx, nbr_idx, c_feat, pooled):
        B, n_src, _ = xyz.shape
        m, ns = nbr_idx.shape[1], nbr_idx.shape[2]
        groups = B * m
        out = torch.empty((groups if pooled else groups * ns, self.n_last), dtype=torch.float32, device=xyz.device)
        call("pa_mlp_chain", 1, 1 if pooled else 0, *self._common(), groups, self.k0, None, 0,
             ptr(xyz), ptr(feat), ptr(center_idx), ptr(nbr_idx), n_src, m, ns, c_feat,
             None, None, None, None, 0, 0, 0, 0, ptr(out), self.n_last)
        return out

    def fp(self, known_feat, idx3, w3, skip, B, n_unknown, m_known, c2, c1):
        rows = B * n_unknown
        out = torch.empty((rows, self.n_last), dtype=torch.float32, device=known_feat.device)
        call("pa_mlp_chain", 2, 0, *self._common(), rows, self.k0, None, 0,
             None, None, None, None, 0, 0, 0, 0,
             ptr(known_feat), ptr(idx3), ptr(w3), ptr(skip), n_unknown, m_known, c2, c1, ptr(out), self.n_last)
        return out

    def plain(self, x):
        rows, k = x.shape
        out = torch.empty((rows, self.n_last), dtype=torch.float32, device=x.device)
        call("pa_mlp_chain", 0, 0, *self._common(), rows, self.k0, ptr(x), k,
             None, None, None, None, 0, 0, 0, 0, None, None, None, None, 0, 0, 0, 0, ptr(out), self.n_last)
        return out


def fold_bn1d(bn):
    """BatchNorm1d (eval) as y*scale + shift, fp64 math."""
    scale = bn.weight.detach().double() / torch.sqrt(bn.running_var.detach().double() + bn.eps)
    shift = bn.bias.detach().double() - bn.running_mean.detach().double() * scale
    return scale, shift


class _Vlad:
    """One pyramid scale for pa_netvlad: BatchNorm folded into the K-major assignment weights."""

    def __init__(self, v, device):
        c, k = v.cluster_weights.shape
        self.c, self.k, self.n = c, k, v.max_samples
        kp = (k + 15) // 16 * 16
        scale, shift = fold_bn1d(v.bn1)
        wc = torch.zeros(c, kp, dtype=torch.float64)
        wc[:, :k] = v.cluster_weights.detach().double().cpu() * scale.cpu()[None, :]
        bias = torch.zeros(kp, dtype=torch.float64)
        bias[:k] = shift.cpu()
        self.wc_t = wc.float().contiguous().to(device)
        self.bias = bias.float().contiguous().to(device)
        self.w2 = v.cluster_weights2.detach()[0].float().contiguous().to(device)      # (C, K)

    def run(self, x, out, ldo, koff):
        b = x.shape[0]
        nfl = _lib.lib().pa_netvlad_scratch_floats(b, self.n, self.k)
        scratch = torch.empty(nfl, dtype=torch.float32, device=x.device)
        call("pa_netvlad", b, self.n, self.c, self.k, ptr(x), ptr(self.wc_t), ptr(self.bias), ptr(self.w2), ptr(scratch),
             ptr(out), ldo, koff)


class _Afa:
    """AdaptiveFeatureAggregator for pa_afa: K-major FC weight, BatchNorm folded to scale/shift."""

    def __init__(self, afa, device):
        if len(afa.mlpa.mlps) != 1:
            raise ValueError("fused APFA supports the single-conv attention layer of the shipped configuration")
        self.watt = afa.mlpa.mlps[0].weight.detach().squeeze(-1).float().contiguous().to(device)   # (out, in)
        self.fc_wt = afa.fc.weight.detach().t().float().contiguous().to(device)                    # (C*K, nout) K-major
        self.fc_bias = afa.fc.bias.detach().float().contiguous().to(device)
        scale, shift = fold_bn1d(afa.bn)
        self.scale, self.shift = scale.float().contiguous().to(device), shift.float().contiguous().to(device)
        self.l2 = 1 if afa.l2_norm else 0
        self.nout = afa.fc.out_features

    def run(self, v):
        b, c, ktot = v.shape
        nfl = _lib.lib().pa_afa_scratch_floats(b, c, ktot, self.nout)
        scratch = torch.empty(nfl, dtype=torch.float32, device=v.device)
        desc = torch.empty((b, self.nout), dtype=torch.float32, device=v.device)
        call("pa_afa", b, c, ktot, self.nout, ptr(v), ptr(self.watt), ptr(self.fc_wt), ptr(self.fc_bias), ptr(self.scale),
             ptr(self.shift), self.l2, ptr(scratch), ptr(desc))
        return desc


class PatchAugNetEngine:
    def __init__(self, model, device):
        self.device = torch.device(device)
        cfg = model.param
        self.sampling, self.knn = list(cfg["SAMPLING"]), list(cfg["KNN"])
        self.use_origin = cfg["USE_ORIGIN_PC_IN_FP"]
        bb = model.backbone
        with torch.no_grad():
            self.sa = [_Chain(fold_shared_mlp(m.mlps[0], self.device)) for m in bb.SA_modules]
            self.fp = [_Chain(fold_shared_mlp(m.mlp, self.device)) for m in bb.FP_modules]
        self.agg = model.aggregation
        agg = self.agg
        self.fused_head = (agg.aggregation_type == 2 and not agg.gating and all(v.feature_size == 256 and v.cluster_size <= 64 for v in agg.vlads)
                           and sum(v.cluster_size for v in agg.vlads) <= 256 and agg.afa.fc.out_features % 16 == 0)
        if self.fused_head:
            with torch.no_grad():
                self.vlads = [_Vlad(v, self.device) for v in agg.vlads]
                self.afa = _Afa(agg.afa, self.device)
        self._key = self._params_key(model)
        self.timer = None     # optional profiling.StageTimer: per-stage HIP-event marks (bench.py kernel attribution)

    def _mark(self, name):
        if self.timer is not None:
            self.timer.mark(name)

    @staticmethod
    def _params_key(model):
        p = next(model.parameters())
        return (p.device, p.data_ptr(), p._version)

    def matches(self, model, x):
        return x.device == self.device and self._key == self._params_key(model)

    def backbone(self, xyz):
        """xyz (B, N, 3) -> point-major features per level + level-0 centre indices."""
        B = xyz.shape[0]
        l_xyz, l_feat, l_c, c_feat = [xyz], [xyz], [], 3
        for i, chain in enumerate(self.sa):
            src = l_xyz[i]
            n, m, ns = src.shape[1], self.sampling[i], self.knn[i]
            cidx = torch.empty((B, m), dtype=torch.int32, device=self.device)
            new_xyz = torch.empty((B, m, 3), dtype=torch.float32, device=self.device)
            call("pa_furthestsampling_gather", B, n, m, ptr(src), ptr(cidx), ptr(new_xyz))
            self._mark(f"sa{i}.fps")
            nbr = torch.empty((B, m, ns), dtype=torch.int32, device=self.device)
            d2 = torch.empty((B, m, ns), dtype=torch.float32, device=self.device)
            call("pa_knnquery", B, n, m, ns, ptr(src), ptr(new_xyz), ptr(nbr), ptr(d2))
            self._mark(f"sa{i}.knn")
            feat = l_feat[i]
            if chain.hidden_ok_pooled:
                y = chain.sa(src, feat, cidx, nbr, c_feat, pooled=True)                      # (B*m, C')
            else:  # wide hidden layers: rows in group order, then the max over each group's ns rows
                full = chain.sa(src, feat, cidx, nbr, c_feat, pooled=False)                  # (B*m*ns, C')
                y = torch.empty((B * m, chain.n_last), dtype=torch.float32, device=self.device)
                call("pa_rowgroup_max", B * m, ns, chain.n_last, ptr(full), ptr(y))
            self._mark(f"sa{i}.chain")
            l_xyz.append(new_xyz)
            l_feat.append(y.view(B, m, chain.n_last))
            l_c.append(cidx)
            c_feat = chain.n_last
        nfp = len(self.fp)
        for i in range(-1, -(nfp + 1), -1):
            chain = self.fp[nfp + i]
            unknown, known = l_xyz[i - 1], l_xyz[i]
            n_u, m_k = unknown.shape[1], known.shape[1]
            w3 = torch.empty((B, n_u, 3), dtype=torch.float32, device=self.device)
            idx3 = torch.empty((B, n_u, 3), dtype=torch.int32, device=self.device)
            call("pa_three_nn_weights", B, n_u, m_k, ptr(unknown), ptr(known), ptr(w3), ptr(idx3))   # patch_aug_net.py:350-353
            self._mark(f"fp{nfp + i}.3nn")
            skip = l_feat[i - 1]
            if i == -nfp and not self.use_origin:
                skip = None
            known_feat = l_feat[i]
            c2 = known_feat.shape[-1]
            c1 = skip.shape[-1] if skip is not None else 0
            y = chain.fp(known_feat.contiguous(), idx3, w3, skip.contiguous() if skip is not None else None, B, n_u, m_k, c2, c1)
            self._mark(f"fp{nfp + i}.chain")
            l_feat[i - 1] = y.view(B, n_u, chain.n_last)
        return l_feat, l_c

    @staticmethod
    def _vlad(v, x):
        """loupe.py:191-222 on point-major x (B, N, C) -- exactly the layout the reference transposes into."""
        act = torch.matmul(x, v.cluster_weights)
        act = F.batch_norm(act.view(-1, v.cluster_size), v.bn1.running_mean, v.bn1.running_var, v.bn1.weight, v.bn1.bias,
                           False, 0.0, v.bn1.eps).view(x.shape[0], -1, v.cluster_size)
        act = torch.softmax(act, dim=-1)
        a = act.sum(-2, keepdim=True) * v.cluster_weights2
        vlad = torch.matmul(act.transpose(1, 2), x).transpose(1, 2) - a
        return F.normalize(vlad, dim=1, p=2)

    def forward(self, x):
        xyz = x.squeeze(1).contiguous()
        self._mark("start")
        l_feat, l_c = self.backbone(xyz)
        nfp = len(self.fp)
        feats = [l_feat[j] for j in range(nfp - 1, -1, -1)]                                   # coarse -> fine, (B, N_i, 256)
        agg = self.agg
        if self.fused_head and x.shape[0] <= 64:
            ktot = sum(v.k for v in self.vlads)
            v = torch.empty((x.shape[0], 256, ktot), dtype=torch.float32, device=self.device)
            koff = 0
            for vl, f in zip(self.vlads, feats):
                vl.run(f.contiguous(), v, ktot, koff)
                koff += vl.k
            self._mark("vlad")
            desc = self.afa.run(v)
            self._mark("afa")
            return desc, self._views(feats, l_c)
        v = torch.cat([self._vlad(vl, f) for vl, f in zip(agg.vlads, feats)], dim=-1)       # (B, 256, sum K)
        self._mark("vlad")
        if agg.aggregation_type == 2:
            desc = agg.afa(v).squeeze(-1)
        elif agg.aggregation_type == 0:
            desc = F.normalize(agg.bn(torch.matmul(v.flatten(1), agg.hidden_weights)))
        else:
            desc = F.normalize(v.max(dim=2)[0])
        if agg.gating:
            desc = agg.context_gating(desc)
        self._mark("afa")
        return desc, self._views(feats, l_c)

    @staticmethod
    def _views(feats, l_c):
        c_o = [l_c[0]]
        for i in range(1, len(l_c)):
            c_o.append(torch.gather(c_o[i - 1], -1, l_c[i].long()))
        return [f.transpose(1, 2).unsqueeze(-1) for f in feats], c_o                         # (B, 256, N_i, 1) views
