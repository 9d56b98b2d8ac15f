"""Aggregation heads: NetVLAD pyramid, adaptive pyramid feature aggregator (APFA), context gating.

Module trees and parameter names follow the reference so that its checkpoints load unchanged
(SURVEY.md appendix B): ``place_recognition/patch_aug_net/models/loupe.py`` (PatchAugNet flavour:
``vlads.{i}``, ``afa``) and ``place_recognition/pptnet_origin/models/loupe.py`` (PPT-Net flavour:
``vlad{i}``, ``hidden_weights``, ``bn2``, ``context_gating``).  Parameters the reference declares but
never uses in forward (``hidden1_weights``, ``bn2``, per-scale ``context_gating``, ``mlpa.trans_conv``,
``mlpa.after_norm``) are kept for state-dict compatibility.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import train_ops
from .backbone import SALayer


def _hip(x):
    """On the MI355X -- train() or eval(), with or without autograd -- the dense layers run on the MFMA GEMM kernels of csrc/train_gemm.hip
    (patchaugnet_amd/train_ops.py); the plain torch statements below are the modules' CPU form."""
    return train_ops.on_device(x)

GroupSALayer = SALayer      # patch_aug_net/models/loupe.py:69-114 -- the same grouped self-attention as pptnet.py's SA_Layer (disabled in
                            # the shipped PatchAugNet config, patch_aug_net.py:239; kept so code that names it keeps importing)


class GatingContext(nn.Module):
    """loupe.py:332-361 -- x * sigmoid(BN(x @ W))."""

    def __init__(self, dim, add_batch_norm=True):
        super().__init__()
        assert add_batch_norm
        self.dim = dim
        self.gating_weights = nn.Parameter(torch.randn(dim, dim) / math.sqrt(dim))
        self.bn1 = nn.BatchNorm1d(dim)

    def forward(self, x):
        if _hip(x) and x.dim() == 2:
            return x * torch.sigmoid(train_ops.bn_rows(self.bn1, train_ops.matmul_rows(x, self.gating_weights), self.bn1.training))
        return x * torch.sigmoid(self.bn1(torch.matmul(x, self.gating_weights)))


class NetVLADBase(nn.Module):
    """loupe.py:159-222 -- soft-assignment VLAD of N local features to K clusters, intra-normalised.

    ``flatten=True`` is the PPT-Net copy (pptnet_origin/models/loupe.py:39-71), which returns the
    (B, C, K) tensor flattened C-major to (B, C*K)."""

    def __init__(self, feature_size, max_samples, cluster_size, output_dim, gating=True, add_batch_norm=True, flatten=False):
        super().__init__()
        assert add_batch_norm
        self.feature_size, self.max_samples, self.cluster_size, self.output_dim = feature_size, max_samples, cluster_size, output_dim
        self.flatten = flatten
        s = 1 / math.sqrt(feature_size)
        self.cluster_weights = nn.Parameter(torch.randn(feature_size, cluster_size) * s)
        self.cluster_weights2 = nn.Parameter(torch.randn(1, feature_size, cluster_size) * s)
        self.hidden1_weights = nn.Parameter(torch.randn(feature_size * cluster_size, output_dim) * s)   # unused in forward
        self.bn1 = nn.BatchNorm1d(cluster_size)
        self.bn2 = nn.BatchNorm1d(output_dim)                                                            # unused in forward
        if gating:
            self.context_gating = GatingContext(output_dim)                                              # unused in forward

    def forward(self, x):
        """x: (B, C, N, 1) -> (B, C, K) (or (B, C*K))."""
        if _hip(x):
            return self._forward_hip(x.squeeze(-1))
        x = x.squeeze(-1).transpose(1, 2)                                  # (B, N, C) view
        act = torch.matmul(x, self.cluster_weights)                        # (B, N, K)
        act = self.bn1(act.reshape(-1, self.cluster_size)).view(-1, self.max_samples, self.cluster_size)
        act = torch.softmax(act, dim=-1)
        a = act.sum(-2, keepdim=True) * self.cluster_weights2              # (B, C, K)
        vlad = torch.matmul(act.transpose(1, 2), x).transpose(1, 2) - a    # (B, C, K)
        vlad = F.normalize(vlad, dim=1, p=2).contiguous()
        return vlad.view(-1, self.cluster_size * self.feature_size) if self.flatten else vlad


def _netvlad_hip(self, x):
    """Channel-major statement of NetVLADBase.forward on the MI355X: x (B, C, N).  The assignment GEMM + BatchNorm is a one-layer chain
    (cluster-major (B, K, N) output, statistics fused into the GEMM epilogue), the aggregation X . act^T a batched k-contiguous GEMM."""
    layer = train_ops.BNLayer(self.cluster_weights, self.bn1, relu=False, transposed=True)
    # assignment GEMM + BatchNorm -> (B, K, N) logits; soft-max, X . act^T - a_sum * cw2, intra-normalisation -> (B, C, K); one autograd node
    vlad = train_ops.netvlad_fused(x, layer, self.cluster_weights2, training=self.training)
    return vlad.view(-1, self.cluster_size * self.feature_size) if self.flatten else vlad


NetVLADBase._forward_hip = _netvlad_hip


class MLPAttentionLayer(nn.Module):
    """loupe.py:8-41 -- per-column attention weight = softmax over columns of max over channels of conv(x)."""

    def __init__(self, channels):
        super().__init__()
        self.mlps = nn.ModuleList(nn.Conv1d(channels[i], channels[i + 1], 1, bias=False) for i in range(len(channels) - 1))
        self.trans_conv = nn.Conv1d(channels[-1], channels[-1], 1)   # unused in forward
        self.after_norm = nn.BatchNorm1d(channels[-1])               # unused in forward

    def forward(self, x):
        r = x
        for mlp in self.mlps:
            r = train_ops.linear_cm(r, mlp.weight) if _hip(x) else torch.matmul(mlp.weight.squeeze(-1), r)
        if _hip(x):
            return train_ops.afa_attention(x, r)                     # the two statements below in one launch (csrc/train_glue.hip)
        w = torch.softmax(r.max(dim=1)[0], dim=-1).unsqueeze(1)      # (B, 1, K)
        return F.relu(x + x * w)


class AdaptiveFeatureAggregator(nn.Module):
    """loupe.py:44-66 -- (B, C, K) -> attention re-weighting -> FC(C*K -> C_out) -> BN -> L2 normalise -> (B, C_out, 1)."""

    def __init__(self, C_in, K, C_out, l2_norm=True):
        super().__init__()
        self.mlpa = MLPAttentionLayer([C_in, C_in])
        self.fc = nn.Linear(C_in * K, C_out)
        self.bn = nn.BatchNorm1d(C_out)
        self.l2_norm = l2_norm

    def forward(self, x):
        x = self.mlpa(x)
        if _hip(x):
            x = train_ops.bn_rows(self.bn, train_ops.linear_rows(x.flatten(1), self.fc.weight, self.fc.bias), self.bn.training)
        else:
            x = self.bn(self.fc(x.flatten(1)))
        if self.l2_norm:
            x = train_ops.l2_normalize(x) if _hip(x) else F.normalize(x)
        return x.unsqueeze(-1)


class SpatialPyramidNetVLAD(nn.Module):
    """PatchAugNet pyramid (loupe.py:225-329).  Aggregation types: 0 FC, 2 APFA across scales and regions
    (the shipped configuration, configs/patch_aug_net.yaml:9), 3 max-pool."""

    def __init__(self, feature_size, max_samples, cluster_size, output_dim, gating=True, aggregation_type=2, add_batch_norm=True):
        super().__init__()
        assert len(feature_size) == len(max_samples) == len(cluster_size) == len(output_dim)
        if aggregation_type not in (0, 2, 3):
            raise NotImplementedError("aggregation_type %r: only 0 (FC), 2 (APFA, shipped) and 3 (max-pool) are built" % aggregation_type)
        self.vlads = nn.ModuleList(NetVLADBase(f, n, k, o, gating, add_batch_norm)
                                   for f, n, k, o in zip(feature_size, max_samples, cluster_size, output_dim))
        k_sum = sum(cluster_size)
        self.gating = gating
        if gating:
            self.context_gating = GatingContext(output_dim[0])
        self.aggregation_type = aggregation_type
        if aggregation_type == 0:
            self.hidden_weights = nn.Parameter(torch.randn(feature_size[0] * k_sum, output_dim[0]) / math.sqrt(feature_size[0]))
            self.bn = nn.BatchNorm1d(output_dim[0])
        elif aggregation_type == 2:
            self.afa = AdaptiveFeatureAggregator(output_dim[0], k_sum, output_dim[0])

    def forward(self, features):
        v = torch.cat([vlad(f) for vlad, f in zip(self.vlads, features)], dim=-1)   # (B, C, sum K)
        if self.aggregation_type == 0 and _hip(v):
            out = train_ops.l2_normalize(train_ops.bn_rows(self.bn, train_ops.matmul_rows(v.flatten(1), self.hidden_weights), self.bn.training))
        elif self.aggregation_type == 0:
            out = F.normalize(self.bn(torch.matmul(v.flatten(1), self.hidden_weights)))
        elif self.aggregation_type == 2:
            out = self.afa(v).squeeze(-1)
        else:
            out = F.normalize(v.max(dim=2)[0])
        return self.context_gating(out) if self.gating else out


class SpatialPyramidNetVLAD4(nn.Module):
    """PPT-Net pyramid (pptnet_origin/models/loupe.py:73-105): four scales, flat concat -> FC -> BN -> gating."""

    def __init__(self, feature_size, max_samples, cluster_size, output_dim, gating=True, add_batch_norm=True):
        super().__init__()
        for i in range(4):
            setattr(self, f"vlad{i}", NetVLADBase(feature_size[i], max_samples[i], cluster_size[i], output_dim[i], gating,
                                                  add_batch_norm, flatten=True))
        self.hidden_weights = nn.Parameter(torch.randn(feature_size[0] * sum(cluster_size), output_dim[0]) / math.sqrt(feature_size[0]))
        self.bn2 = nn.BatchNorm1d(output_dim[0])
        self.gating = gating
        if gating:
            self.context_gating = GatingContext(output_dim[0])

    def forward(self, f0, f1, f2, f3):
        v = torch.cat([self.vlad0(f0), self.vlad1(f1), self.vlad2(f2), self.vlad3(f3)], dim=-1)
        if _hip(v):
            v = train_ops.bn_rows(self.bn2, train_ops.matmul_rows(v, self.hidden_weights), self.bn2.training)
        else:
            v = self.bn2(torch.matmul(v, self.hidden_weights))
        return self.context_gating(v) if self.gating else v
