"""PointNetVLAD (BASELINE.json configs[0]: the reference's CPU-runnable model; plumbing, no native ops).

Model API and state-dict keys of ``place_recognition/pointnet_vlad/PointNetVlad.py:235-247`` as built by
``place_recognition/evaluate.py:88-90``:
``PointNetVlad(global_feat=True, feature_transform=True, max_pool=False, output_dim=256, num_points=4096)``;
``forward(x: (B,1,N,3)) -> (B, output_dim)``.  The reference has no CUDA extension on this path -- it is dense torch ops only
-- so the CPU form below is torch as well (1x1 convolutions written as matmuls on point-major activations).  evaluate.py moves the model
to the accelerator (``model.to(device)``), so a tensor on the MI355X is served too: the ``_hip`` methods state the same network
channel-major on the hand-written MFMA GEMM / BatchNorm / NetVLAD kernels of ``train_ops`` (csrc/train_gemm.hip, train_glue.hip) -- the
kernels PatchAugNet's module path trains on -- in train() and eval(), with autograd; no rocBLAS / MIOpen kernel is reached
(tests/test_pointnet_vlad.py).  Keys: tests/golden/pointnet_vlad_state_dict_keys.json.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import train_ops
from .loupe import GatingContext

__all__ = ["PointNetVlad"]


def _pointwise(conv, bn, x, relu=True):
    """Conv2d with a (1, k) kernel on (B, N, C_in) point-major rows == one matmul; BatchNorm2d over channels."""
    y = x @ conv.weight.flatten(1).t() + conv.bias
    if bn is not None:
        y = F.batch_norm(y.transpose(1, 2), bn.running_mean, bn.running_var, bn.weight, bn.bias, bn.training, bn.momentum, bn.eps).transpose(1, 2)
    return F.relu(y) if relu else y


class STN3d(nn.Module):
    """Spatial transformer (PointNetVlad.py:122-178): k = 3 on raw xyz, k = 64 on features; fc3 starts at zero."""

    def __init__(self, num_points=2500, k=3, use_bn=True):
        super().__init__()
        self.k, self.use_bn = k, use_bn
        self.conv1 = nn.Conv2d(1 if k == 3 else k, 64, (1, 3 if k == 3 else 1))
        self.conv2 = nn.Conv2d(64, 128, (1, 1))
        self.conv3 = nn.Conv2d(128, 1024, (1, 1))
        self.fc1, self.fc2, self.fc3 = nn.Linear(1024, 512), nn.Linear(512, 256), nn.Linear(256, k * k)
        nn.init.zeros_(self.fc3.weight)
        nn.init.zeros_(self.fc3.bias)
        if use_bn:
            self.bn1, self.bn2, self.bn3 = nn.BatchNorm2d(64), nn.BatchNorm2d(128), nn.BatchNorm2d(1024)
            self.bn4, self.bn5 = nn.BatchNorm1d(512), nn.BatchNorm1d(256)

    def forward(self, x):
        """x: (B, N, k) point-major -> (B, k, k)."""
        bn = (lambda i: getattr(self, f"bn{i}")) if self.use_bn else (lambda i: None)
        x = _pointwise(self.conv1, bn(1), x)
        x = _pointwise(self.conv2, bn(2), x)
        x = _pointwise(self.conv3, bn(3), x)
        x = x.max(dim=1)[0]                                           # MaxPool2d((num_points, 1))
        x = self.fc1(x)
        x = F.relu(self.bn4(x) if self.use_bn else x)
        x = self.fc2(x)
        x = F.relu(self.bn5(x) if self.use_bn else x)
        x = self.fc3(x) + torch.eye(self.k, dtype=x.dtype, device=x.device).flatten()
        return x.view(-1, self.k, self.k)


def _conv_cm(conv, bn, x, training, relu=True):
    """The same layer on channel-major (B, C_in, N) device rows: conv weight (O, 1, 1, C_in) or (O, C_in, 1, 1) is an (O, C_in) matrix either way."""
    w = conv.weight.flatten(1)
    if bn is not None:
        return train_ops.chain_train(x, [train_ops.BNLayer(w, bn, bias=conv.bias, relu=relu)], training=training)
    y = train_ops.linear_cm(x, w, conv.bias)
    return F.relu(y) if relu else y


def _stn_hip(self, x):
    """STN3d.forward on the MI355X: x (B, k, N) channel-major -> (B, k, k)."""
    tr = self.training
    bn = (lambda i: getattr(self, f"bn{i}")) if self.use_bn else (lambda i: None)
    x = _conv_cm(self.conv1, bn(1), x, tr)
    x = _conv_cm(self.conv2, bn(2), x, tr)
    x = _conv_cm(self.conv3, bn(3), x, tr)
    x = x.max(dim=2)[0]                                               # (B, 1024)
    x = train_ops.linear_rows(x, self.fc1.weight, self.fc1.bias)
    x = F.relu(train_ops.bn_rows(self.bn4, x, self.bn4.training) if self.use_bn else x)
    x = train_ops.linear_rows(x, self.fc2.weight, self.fc2.bias)
    x = F.relu(train_ops.bn_rows(self.bn5, x, self.bn5.training) if self.use_bn else x)
    x = train_ops.linear_rows(x, self.fc3.weight, self.fc3.bias) + torch.eye(self.k, dtype=x.dtype, device=x.device).flatten()
    return x.view(-1, self.k, self.k)


STN3d._forward_hip = _stn_hip


class PointNetfeat(nn.Module):
    """PointNetVlad.py:181-232."""

    def __init__(self, num_points=2500, global_feat=True, feature_transform=False, max_pool=True):
        super().__init__()
        self.stn = STN3d(num_points=num_points, k=3, use_bn=False)
        self.feature_trans = STN3d(num_points=num_points, k=64, use_bn=False)
        self.apply_feature_trans = feature_transform
        self.conv1 = nn.Conv2d(1, 64, (1, 3))
        self.conv2, self.conv3 = nn.Conv2d(64, 64, (1, 1)), nn.Conv2d(64, 64, (1, 1))
        self.conv4, self.conv5 = nn.Conv2d(64, 128, (1, 1)), nn.Conv2d(128, 1024, (1, 1))
        self.bn1, self.bn2, self.bn3 = nn.BatchNorm2d(64), nn.BatchNorm2d(64), nn.BatchNorm2d(64)
        self.bn4, self.bn5 = nn.BatchNorm2d(128), nn.BatchNorm2d(1024)
        self.num_points, self.global_feat, self.max_pool = num_points, global_feat, max_pool

    def forward(self, x):
        """x: (B, 1, N, 3) -> (B, 1024, N, 1) when max_pool is False (the evaluate.py configuration)."""
        if train_ops.on_device(x):
            return self._forward_hip(x)
        p = x.squeeze(1)
        trans = self.stn(p)
        p = torch.matmul(p, trans)
        f = _pointwise(self.conv1, self.bn1, p)
        f = _pointwise(self.conv2, self.bn2, f)
        pointfeat = f
        if self.apply_feature_trans:
            f = torch.matmul(f, self.feature_trans(f))
        f = _pointwise(self.conv3, self.bn3, f)
        f = _pointwise(self.conv4, self.bn4, f)
        f = _pointwise(self.conv5, self.bn5, f, relu=False)
        if not self.max_pool:
            return f.transpose(1, 2).unsqueeze(-1)
        g = f.max(dim=1)[0]
        if self.global_feat:
            return g, trans
        return torch.cat([g.unsqueeze(-1).expand(-1, -1, self.num_points), pointfeat.transpose(1, 2)], 1), trans


def _feat_hip(self, x):
    """PointNetfeat.forward on the MI355X, channel-major: p (B, 3, N); `p @ trans` of the point-major form is trans^T . p here."""
    tr = self.training
    p = x.squeeze(1).transpose(1, 2).contiguous()                     # (B, 3, N)
    trans = self.stn._forward_hip(p)
    p = train_ops.bmm_nn(trans.transpose(1, 2), p)
    f = _conv_cm(self.conv1, self.bn1, p, tr)
    f = _conv_cm(self.conv2, self.bn2, f, tr)
    pointfeat = f
    if self.apply_feature_trans:
        f = train_ops.bmm_nn(self.feature_trans._forward_hip(f).transpose(1, 2), f)
    f = _conv_cm(self.conv3, self.bn3, f, tr)
    f = _conv_cm(self.conv4, self.bn4, f, tr)
    f = _conv_cm(self.conv5, self.bn5, f, tr, relu=False)             # (B, 1024, N)
    if not self.max_pool:
        return f.unsqueeze(-1)
    g = f.max(dim=2)[0]
    if self.global_feat:
        return g, trans
    return torch.cat([g.unsqueeze(-1).expand(-1, -1, self.num_points), pointfeat], 1), trans


PointNetfeat._forward_hip = _feat_hip


class NetVLADLoupe(nn.Module):
    """PointNetVlad.py:12-79 -- NetVLAD with the flat C-major layout, a second L2 normalisation, FC, BN and gating."""

    def __init__(self, feature_size, max_samples, cluster_size, output_dim, gating=True, add_batch_norm=True, is_training=True):
        super().__init__()
        assert add_batch_norm
        self.feature_size, self.max_samples, self.cluster_size, self.output_dim = feature_size, max_samples, cluster_size, output_dim
        s = 1 / math.sqrt(feature_size)
        self.cluster_weights = nn.Parameter(torch.randn(feature_size, cluster_size) * s)
        self.cluster_weights2 = nn.Parameter(torch.randn(1, feature_size, cluster_size) * s)
        self.hidden1_weights = nn.Parameter(torch.randn(cluster_size * feature_size, output_dim) * s)
        self.bn1 = nn.BatchNorm1d(cluster_size)
        self.bn2 = nn.BatchNorm1d(output_dim)
        self.gating = gating
        if gating:
            self.context_gating = GatingContext(output_dim, add_batch_norm=True)

    def forward(self, x):
        """x: (B, C, N, 1)."""
        if train_ops.on_device(x):
            return self._forward_hip(x.squeeze(-1))
        x = x.squeeze(-1).transpose(1, 2)                                           # (B, N, C)
        act = torch.matmul(x, self.cluster_weights)
        act = self.bn1(act.reshape(-1, self.cluster_size)).view(-1, self.max_samples, self.cluster_size)
        act = torch.softmax(act, dim=-1)
        a = act.sum(-2, keepdim=True) * self.cluster_weights2
        vlad = torch.matmul(act.transpose(1, 2), x).transpose(1, 2) - a             # (B, C, K)
        vlad = F.normalize(vlad, dim=1, p=2).reshape(-1, self.cluster_size * self.feature_size)
        vlad = F.normalize(vlad, dim=1, p=2)
        vlad = self.bn2(torch.matmul(vlad, self.hidden1_weights))
        return self.context_gating(vlad) if self.gating else vlad


def _loupe_hip(self, x):
    """NetVLADLoupe.forward on the MI355X: x (B, C, N) channel-major (the layout loupe.NetVLADBase's device form uses)."""
    tr = self.training
    pre = train_ops.chain_train(x, [train_ops.BNLayer(self.cluster_weights, self.bn1, relu=False, transposed=True)], training=tr)   # (B, K, N)
    vlad = train_ops.netvlad_tail(pre, x, self.cluster_weights2)                    # soft-max, X . act^T - a_sum * cw2, normalise over C: (B, C, K)
    vlad = train_ops.l2_normalize(vlad.reshape(-1, self.cluster_size * self.feature_size))
    vlad = train_ops.bn_rows(self.bn2, train_ops.matmul_rows(vlad, self.hidden1_weights), self.bn2.training)
    return self.context_gating(vlad) if self.gating else vlad


NetVLADLoupe._forward_hip = _loupe_hip


class PointNetVlad(nn.Module):
    def __init__(self, num_points=2500, global_feat=True, feature_transform=False, max_pool=True, output_dim=1024):
        super().__init__()
        self.point_net = PointNetfeat(num_points=num_points, global_feat=global_feat, feature_transform=feature_transform, max_pool=max_pool)
        self.net_vlad = NetVLADLoupe(feature_size=1024, max_samples=num_points, cluster_size=64, output_dim=output_dim,
                                     gating=True, add_batch_norm=True, is_training=True)

    def forward(self, x):
        return self.net_vlad(self.point_net(x))
