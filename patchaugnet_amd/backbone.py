"""PointNet++-style pyramid backbone shared by PatchAugNet and PPT-Net (module path).

Set-abstraction (SA) level  = FPS -> gather centres -> kNN grouping with centre subtraction (EdgeConv style)
                              -> shared MLP -> max over the neighbourhood [-> grouped self-attention, PPT-Net only];
feature-propagation (FP)    = 3-NN inverse-distance interpolation -> concat skip features -> shared MLP.

Module trees / parameter names follow ``place_recognition/patch_aug_net/models/patch_aug_net.py:110-363`` and
``place_recognition/pptnet_origin/models/pptnet.py:65-340`` so reference checkpoints load unchanged.  This is the
autograd-capable path (training, and the reference-shaped intermediate tensors); evaluation runs the fused HIP
engine instead (patchaugnet_amd/engine.py), which reads the very same parameters.
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import pointops, train_ops
from .pt_util import SharedMLP


class SALayer(nn.Module):
    """Grouped self-attention of PPT-Net (pptnet.py:246-282): q and k share one grouped 1x1 conv, so the energy is
    Y^T Y; row softmax, then column re-normalisation; x + relu(BN(conv(x - x_v @ attn)))."""

    def __init__(self, channels, gp):
        super().__init__()
        assert channels % 4 == 0
        self.gp = gp
        self.q_conv = nn.Conv1d(channels, channels, 1, bias=False, groups=gp)
        self.k_conv = nn.Conv1d(channels, channels, 1, bias=False, groups=gp)
        self.q_conv.weight = self.k_conv.weight          # tied, both names stay in the state-dict (pptnet.py:254)
        self.v_conv = nn.Conv1d(channels, channels, 1)
        self.trans_conv = nn.Conv1d(channels, channels, 1)
        self.after_norm = nn.BatchNorm1d(channels)

    def _fused(self, x):
        """eval + no_grad on the MI355X: pa_linear + pa_sa_attention + pa_linear (csrc/attention.hip), no (B, gp, N, N) tensor."""
        from .engine import _Attn
        ts = (self.k_conv.weight, self.v_conv.weight, self.v_conv.bias, self.trans_conv.weight, self.trans_conv.bias, self.after_norm.weight,
              self.after_norm.bias, self.after_norm.running_mean, self.after_norm.running_var)
        key = (x.device,) + tuple((t.data_ptr(), t._version) for t in ts)
        with torch.cuda.device(x.device):     # the C ABI launches on the CURRENT device's stream: follow the tensor
            if getattr(self, "_attn_key", None) != key:
                self._attn, self._attn_key = _Attn(self, x.device), key
            bs, ch, n = x.shape
            xm = x.transpose(1, 2).contiguous().view(bs * n, ch)
            return self._attn.run(xm, bs, n).view(bs, n, ch).transpose(1, 2).contiguous()

    def __getstate__(self):
        st = self.__dict__.copy()
        st.pop("_attn", None)                  # folded copies hold device buffers bound to this module's current weights
        st.pop("_attn_key", None)
        return st

    def _autograd_hip(self, x):
        """train(), or eval() with autograd, on the MI355X: the same function on the MFMA GEMM kernels of csrc/train_gemm.hip and the
        soft-max / re-normalisation kernels of csrc/attention_train.hip, forward and backward (patchaugnet_amd/train_ops.py).  The tied
        grouped q / k convolution is evaluated as its block-diagonal dense matrix (gp x the multiply-adds of the grouped form, C <= 512: the
        N x N products dominate); its gradient flows back to the (C, C/gp) parameter through the block selection."""
        from . import train_ops
        w = self.k_conv.weight.squeeze(-1)                                        # (C, C/gp): output o reads the inputs of group o // cg
        cg = w.shape[1]
        dense = torch.block_diag(*w.view(self.gp, cg, cg))                          # (C, C)
        y = train_ops.linear_cm(x, dense)
        x_v = train_ops.linear_cm(x, self.v_conv.weight, self.v_conv.bias)
        x_r = train_ops.sa_attention_train(y, x_v)
        layer = train_ops.BNLayer(self.trans_conv.weight, self.after_norm, bias=self.trans_conv.bias)
        return x + train_ops.chain_train((x - x_r).contiguous(), [layer], training=self.training)

    def forward(self, x):
        if x.is_cuda:
            if not self.training and not torch.is_grad_enabled() and x.shape[1] in (64, 128, 256, 512):
                return self._fused(x)
            return self._autograd_hip(x)
        bs, ch, n = x.shape                         # CPU form
        y = self.k_conv(x).reshape(bs, self.gp, ch // self.gp, n)
        energy = torch.matmul(y.permute(0, 1, 3, 2), y).sum(dim=1)
        attn = torch.softmax(energy, dim=-1)
        attn = attn / (1e-9 + attn.sum(dim=1, keepdim=True))
        x_r = torch.matmul(self.v_conv(x), attn)
        return x + F.relu(self.after_norm(self.trans_conv(x - x_r)))


class SAModuleMSG(nn.Module):
    """Set abstraction with MULTI-SCALE grouping (``PointNet2SAModuleMSG``, patch_aug_net.py:246-287 with the base class's forward, :203-243):
    one sampling of ``npoint`` centres, then per scale i its own EdgeConv grouper (``radii[i]`` / ``nsamples[i]``: ball or dilated kNN), shared MLP
    ``mlps[i]`` and max over the neighbourhood; the scales' features are concatenated along the channels and their neighbour lists along the
    last axis.  Parameter names (``groupers.{i}``, ``mlps.{i}...``) follow the reference.  The shipped configurations use ONE scale per level
    (SAModule below, which the fused engine evaluates); a level with several scales runs here, on the HIP op layer + the MFMA training kernels
    (forward and backward), and the fused engine refuses it."""

    def __init__(self, *, npoint, radii, nsamples, knn_dilation=1, mlps, gp=None, attention=False, use_xyz=True):
        super().__init__()
        assert len(radii) == len(nsamples) == len(mlps) and len(mlps) >= 1
        self.npoint = npoint
        self.groupers = nn.ModuleList()
        self.mlps = nn.ModuleList()
        for radius, nsample, mlp in zip(radii, nsamples, mlps):
            spec = list(mlp)                                   # (the reference adds the 3 to the caller's own list; a copy here)
            if use_xyz:
                spec[0] += 3
            self.groupers.append(pointops.QueryAndGroup_Edge(radius, nsample, knn_dilation=knn_dilation, use_xyz=use_xyz, ret_sample_idx=True)
                                 if npoint is not None else pointops.GroupAll(use_xyz))
            self.mlps.append(SharedMLP(spec, bn=True))
        if attention:                                          # pptnet.py:137-244: one grouped self-attention per scale, on that scale's output
            self.sas = nn.ModuleList(SALayer(m.channels[-1], gp) for m in self.mlps)

    def forward(self, xyz, features, geo=None):
        assert geo is None, "precomputed geometry is built for single-scale levels (SAModule)"
        center_idx = pointops.furthestsampling(xyz, self.npoint)
        new_xyz = pointops.gathering(xyz.transpose(1, 2).contiguous(), center_idx).transpose(1, 2).contiguous()
        center_features = pointops.gathering(features, center_idx)
        feats, idxs = [], []
        for i, (grouper, mlp) in enumerate(zip(self.groupers, self.mlps)):
            grouped, sample_idx = grouper(xyz, new_xyz, features, center_features)         # (B, C, M, K_i)
            y = mlp.forward_maxpool(grouped)                                              # (B, C'_i, M)
            if hasattr(self, "sas"):
                y = self.sas[i](y)
            feats.append(y)
            idxs.append(sample_idx)
        return new_xyz, center_idx, torch.cat(idxs, dim=-1), torch.cat(feats, dim=1)


class SAModule(SAModuleMSG):
    """One set-abstraction level with a single scale (``PointNet2SAModule``, patch_aug_net.py:290-314 / pptnet.py:137-244)."""

    def __init__(self, *, mlp, npoint, nsample, knn_dilation=1, gp=None, attention=False, radius=None, use_xyz=True):
        super().__init__(npoint=npoint, radii=[radius], nsamples=[nsample], knn_dilation=knn_dilation, mlps=[mlp], gp=gp, attention=attention,
                         use_xyz=use_xyz)

    @torch.no_grad()
    def geometry(self, xyz, coordinate_features=None):
        """Everything of this level that depends on the coordinates only: (centre indices, centre coordinates, neighbour indices, grouped coordinates,
        grouped input or None) -- the sampling and the grouper's search / permutation, exactly as forward() computes them.  coordinate_features:
        the level's input features when they are the coordinates themselves (the first level: (B, 3, n) = xyz transposed); the grouper's whole
        output is then coordinate-only work too."""
        center_idx, new_xyz = pointops.furthestsampling_gather(xyz, self.npoint)
        idx = self.groupers[0].neighbours(xyz, new_xyz)
        grouped = None
        fused_first = (coordinate_features is not None and xyz.dtype == torch.float32 and idx.shape[2] > 1 and self.groupers[0].use_xyz
                       and type(self.groupers[0]) is pointops.QueryAndGroup_Edge)
        if fused_first:
            # the features ARE the coordinates: grouped features minus centre features repeat the centred coordinates (same subtraction, same operands)
            o_grouped, grouped = pointops.grouped_coordinates_fused(xyz, new_xyz, idx, 2)
            coords = (o_grouped, grouped[:, :3])
        else:
            coords = pointops.grouped_coordinates(xyz, new_xyz, idx)
            if coordinate_features is not None:
                center_features = pointops.gathering(coordinate_features, center_idx)
                grouped = self.groupers[0](xyz, new_xyz, coordinate_features, center_features, idx=idx, coords=coords)[0]
        return center_idx, new_xyz, idx, coords, grouped

    def forward(self, xyz, features, geo=None):
        if geo is None:
            center_idx, new_xyz = pointops.furthestsampling_gather(xyz, self.npoint)
            idx = coords = grouped = None
        else:
            center_idx, new_xyz, idx, coords, grouped = geo
        if grouped is None:
            # centres by index: gathering + grouping + subtract + cat as one op each way on the device (pointops.EdgeGroup)
            grouped, sample_idx = self.groupers[0](xyz, new_xyz, features, None, idx=idx, coords=coords, center_idx=center_idx)
        else:                       # first level, everything about its input was coordinate-only (geometry(..., coordinate_features=...))
            sample_idx = idx
        y = self.mlps[0].forward_maxpool(grouped)
        if hasattr(self, "sas"):
            y = self.sas[0](y)
        return new_xyz, center_idx, sample_idx, y


class FPModule(nn.Module):
    """Feature propagation (patch_aug_net.py:317-363)."""

    fold_first_layer = os.environ.get("PA_FP_NO_FOLD") is None      # A/B and test switch: False = interpolation -> cat -> SharedMLP as written in the reference

    def __init__(self, *, mlp):
        super().__init__()
        self.mlp = SharedMLP(mlp, bn=True)

    @staticmethod
    @torch.no_grad()
    def geometry(unknown, known, lists_for_channels=0):
        """(3-NN indices, inverse-distance weights, backward lists or None): the coordinate-only part of forward().  lists_for_channels = the
        channel count of the features that will be interpolated under autograd (0 = no backward pass follows)."""
        idx, weight = pointops.three_nn_weights(unknown, known)
        lists = None
        if lists_for_channels and pointops._gather_form(idx.shape[0], lists_for_channels, idx.shape[1], known.shape[1]):
            lists = pointops.interpolation_backward_lists(idx, weight, known.shape[1])      # the backward pass's inverted lists, off the step's path
        return idx, weight, lists

    def forward(self, unknown, known, unknown_feats, known_feats, geo=None):
        if geo is None:
            idx, weight = pointops.three_nn_weights(unknown, known)     # search + inverse-distance weights (patch_aug_net.py:350-353), no gradient
            lists = None
        else:
            idx, weight, lists = geo
        if self.fold_first_layer and train_ops.fp_fold_applies(known_feats, unknown_feats, idx, len(self.mlp)) and self.mlp[0].conv.weight.shape[0] % 8 == 0:
            # the first 1x1 convolution folded through the interpolation: its contractions run on the known points (csrc/fp_fold_train.hip)
            layers = [train_ops.BNLayer(l.conv.weight, l.bn.bn) for l in self.mlp]
            return train_ops.fp_chain_train_folded(known_feats, unknown_feats, idx, weight, lists, layers, training=self.mlp.training)
        x = pointops.interpolation(known_feats, idx, weight, lists)
        if unknown_feats is not None:
            x = torch.cat([x, unknown_feats], dim=1)
        return self.mlp(x.unsqueeze(-1)).squeeze(-1)


def origin_indices(l_center_idx, l_sample_idx):
    """Map level-local centre / neighbour indices back to indices of the input cloud (patch_aug_net.py:169-177)."""
    c_o, s_o = [l_center_idx[0]], [l_sample_idx[0]]
    if all(t.is_cuda and t.dtype == torch.int32 for t in list(l_center_idx) + list(l_sample_idx)):
        for i in range(1, len(l_center_idx)):
            c_o.append(pointops.compose_indices(c_o[i - 1], l_center_idx[i]))
            s_o.append(pointops.compose_indices(c_o[i - 1], l_sample_idx[i]))
        return c_o, s_o
    for i in range(1, len(l_center_idx)):
        c_o.append(torch.gather(c_o[i - 1], -1, l_center_idx[i].long()))
        s_o.append(torch.gather(c_o[i - 1].unsqueeze(1).expand(-1, l_sample_idx[i].shape[1], -1), -1, l_sample_idx[i].long()))
    return c_o, s_o


class PyramidBackbone(nn.Module):
    """SA_modules / FP_modules lists.  ``sa_mlps``/``fp_mlps`` are the channel specs; FP_modules[0] is the finest
    level and the forward pass walks the list backwards, exactly like the reference (patch_aug_net.py:183-187)."""

    def __init__(self, *, sampling, knn, sa_mlps, fp_mlps, knn_dilation=1, gp=8, attention=False, use_origin_pc_in_fp=True):
        super().__init__()
        self.use_origin_pc_in_fp = use_origin_pc_in_fp
        # a level whose spec is a LIST of specs (with a list of nsample) is a multi-scale-grouping level (PointNet2SAModuleMSG)
        self.SA_modules = nn.ModuleList(
            SAModuleMSG(mlps=m, npoint=s, nsamples=list(k), radii=[None] * len(m), knn_dilation=knn_dilation, gp=gp, attention=attention)
            if isinstance(m[0], (list, tuple)) else SAModule(mlp=m, npoint=s, nsample=k, knn_dilation=knn_dilation, gp=gp, attention=attention)
            for m, s, k in zip(sa_mlps, sampling, knn))
        self.FP_modules = nn.ModuleList(FPModule(mlp=m) for m in fp_mlps)

    @torch.no_grad()
    def geometry(self, pointcloud):
        """The coordinate-only part of forward(): sampling, neighbour search (with the groupers' permutation) and 3-NN weights of every level.
        Depends on the input cloud, not on the weights, so a training loop can compute it for the NEXT batch while the current one trains
        (train.GraphedTrainer); forward(pointcloud, geometry=...) then skips those launches.  {"sa": [...], "fp": [...]} of tensors."""
        l_xyz, sa_geo = [pointcloud], []
        for i, sa in enumerate(self.SA_modules):
            g = sa.geometry(l_xyz[-1], coordinate_features=pointcloud.transpose(1, 2).contiguous() if i == 0 else None)
            sa_geo.append(g)
            l_xyz.append(g[1])
        nfp = len(self.FP_modules)
        fp_geo = [None] * nfp
        for i in range(-1, -(nfp + 1), -1):
            # channels of the coarser level's features this level interpolates (its backward lists are coordinate-only work too; train() only)
            c_known = self.SA_modules[-1].mlps[0].channels[-1] if i == -1 else self.FP_modules[i + 1].mlp.channels[-1]
            fp_geo[i] = FPModule.geometry(l_xyz[i - 1], l_xyz[i], lists_for_channels=c_known if self.training else 0)
        # level-local centre / neighbour indices mapped to indices of the input cloud: index arithmetic only
        origin = origin_indices([g[0] for g in sa_geo], [g[2] for g in sa_geo])
        return {"sa": sa_geo, "fp": fp_geo, "origin": origin}

    def forward(self, pointcloud, geometry=None):
        l_xyz, l_feat = [pointcloud], [pointcloud.transpose(1, 2).contiguous()]
        l_c, l_s = [], []
        for i, sa in enumerate(self.SA_modules):
            nx, ci, si, f = sa(l_xyz[i], l_feat[i], geo=None if geometry is None else geometry["sa"][i])
            l_xyz.append(nx); l_feat.append(f); l_c.append(ci); l_s.append(si)
        sa_features = list(l_feat[1:])
        c_o, s_o = origin_indices(l_c, l_s) if geometry is None else geometry["origin"]
        nfp = len(self.FP_modules)
        for i in range(-1, -(nfp + 1), -1):
            skip = l_feat[i - 1]
            if i == -nfp and not self.use_origin_pc_in_fp:
                skip = None
            l_feat[i - 1] = self.FP_modules[i](l_xyz[i - 1], l_xyz[i], skip, l_feat[i], geo=None if geometry is None else geometry["fp"][i])
        fp = [l_feat[j].unsqueeze(-1) for j in range(nfp - 1, -1, -1)]          # coarse -> fine
        return {"center_idx_origin": c_o, "sample_idx_origin": s_o, "sa_features": sa_features, "fp_features": fp,
                "l_xyz": l_xyz,
                # the first level's grouped neighbour coordinates (B, 3, m0, k) when the geometry was precomputed (= grouping(xyz, sample_idx_origin[0]))
                "origin_patches_cm": None if geometry is None else geometry["sa"][0][3][0]}
