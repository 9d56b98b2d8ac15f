"""CPU (oracle-backed) stand-in for the reference's Python op layer.

TEST INFRASTRUCTURE ONLY.  It exposes the public names of
``libs/pointops/functions/pointops.py`` (reference, lines 29-661) on CPU torch
tensors by calling oracle/pointops_oracle.c, so that

  * oracle/gen_golden.py can register it as ``libs.pointops.functions.pointops``
    and run the reference's own, unmodified model classes on CPU (SURVEY.md
    section 8c) to produce tests/golden/ vectors;
  * oracle/models_cpu.py (the CPU restatement of the networks) has an op layer.

The product op layer is patchaugnet_amd/pointops.py, which only accepts device
tensors and calls the HIP library; it never imports this file.
"""
import numpy as np
import torch
import torch.nn as nn
from torch.autograd import Function

from . import oracle_ops as _o


def _np(t):
    return t.detach().cpu().contiguous().numpy()


def _t(a, like=None):
    return torch.from_numpy(np.ascontiguousarray(a))


def furthestsampling(xyz, m):
    """pointops.py:11-29"""
    assert xyz.is_contiguous()
    return _t(_o.furthestsampling(_np(xyz), int(m)))


class _Gathering(Function):
    """pointops.py:32-57"""

    @staticmethod
    def forward(ctx, features, idx):
        assert features.is_contiguous() and idx.is_contiguous()
        ctx.save_for_backward(idx)
        ctx.n = features.shape[2]
        return _t(_o.gathering_forward(_np(features), _np(idx)))

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        return _t(_o.gathering_backward(_np(g), _np(idx), ctx.n)), None


gathering = _Gathering.apply


def nearestneighbor(unknown, known):
    """pointops.py:60-82 -- returns (sqrt(dist2), idx)"""
    assert unknown.is_contiguous() and known.is_contiguous()
    d2, idx = _o.nearestneighbor(_np(unknown), _np(known))
    return torch.sqrt(_t(d2)), _t(idx)


class _Interpolation(Function):
    """pointops.py:85-118"""

    @staticmethod
    def forward(ctx, features, idx, weight):
        assert features.is_contiguous() and idx.is_contiguous() and weight.is_contiguous()
        ctx.save_for_backward(idx, weight)
        ctx.m = features.shape[2]
        return _t(_o.interpolation_forward(_np(features), _np(idx), _np(weight)))

    @staticmethod
    def backward(ctx, g):
        idx, weight = ctx.saved_tensors
        return _t(_o.interpolation_backward(_np(g), _np(idx), _np(weight), ctx.m)), None, None


interpolation = _Interpolation.apply


class _Grouping(Function):
    """pointops.py:121-150"""

    @staticmethod
    def forward(ctx, features, idx):
        assert features.is_contiguous() and idx.is_contiguous()
        ctx.save_for_backward(idx)
        ctx.n = features.shape[2]
        return _t(_o.grouping_forward(_np(features), _np(idx)))

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        return _t(_o.grouping_backward(_np(g), _np(idx), ctx.n)), None


grouping = _Grouping.apply


def grouping_int(features, idx):
    """pointops.py:153-172"""
    return _t(_o.grouping_int_forward(_np(features), _np(idx)))


def ballquery(radius, nsample, xyz, new_xyz):
    """pointops.py:175-197"""
    assert xyz.is_contiguous() and new_xyz.is_contiguous()
    return _t(_o.ballquery(float(radius), int(nsample), _np(xyz), _np(new_xyz)))


def knnquery(nsample, xyz, new_xyz=None):
    """pointops.py:407-433 (dist2 is computed and discarded there too)"""
    if new_xyz is None:
        new_xyz = xyz
    assert xyz.is_contiguous() and new_xyz.is_contiguous()
    idx, _ = _o.knnquery(int(nsample), _np(xyz), _np(new_xyz))
    return _t(idx)


def featuredistribute(max_xyz, xyz):
    """pointops.py:200-221"""
    return _t(_o.featuredistribute(_np(max_xyz), _np(xyz)))


class _FeatureGather(Function):
    """pointops.py:224-256"""

    @staticmethod
    def forward(ctx, max_feature, distribute_idx):
        ctx.save_for_backward(distribute_idx)
        ctx.n = max_feature.shape[2]
        return _t(_o.gathering_forward(_np(max_feature), _np(distribute_idx)))

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        return _t(_o.gathering_backward(_np(g), _np(idx), ctx.n)), None


featuregather = _FeatureGather.apply


def labelstat_ballrange(radius, xyz, new_xyz, label_stat):
    """pointops.py:259-285"""
    return _t(_o.labelstat_ballrange(float(radius), _np(xyz), _np(new_xyz), _np(label_stat)))


def labelstat_idx(nsample, label_stat, idx):
    """pointops.py:288-312"""
    return _t(_o.labelstat_idx(_np(label_stat), _np(idx)))


def labelstat_and_ballquery(radius, nsample, xyz, new_xyz, label_stat):
    """pointops.py:315-344"""
    a, b = _o.labelstat_and_ballquery(float(radius), int(nsample), _np(xyz), _np(new_xyz), _np(label_stat))
    return _t(a), _t(b)


def _query(radius, nsample, xyz, new_xyz):
    return ballquery(radius, nsample, xyz, new_xyz) if radius is not None else knnquery(nsample, xyz, new_xyz)


class QueryAndGroup(nn.Module):
    """pointops.py:476-516"""

    def __init__(self, radius=None, nsample=32, use_xyz=True):
        super().__init__()
        self.radius, self.nsample, self.use_xyz = radius, nsample, use_xyz

    def forward(self, xyz, new_xyz=None, features=None, idx=None):
        new_xyz = xyz if new_xyz is None else new_xyz
        if idx is None:
            idx = _query(self.radius, self.nsample, xyz, new_xyz)
        g_xyz = grouping(xyz.transpose(1, 2).contiguous(), idx) - new_xyz.transpose(1, 2).unsqueeze(-1)
        if features is None:
            assert self.use_xyz
            return g_xyz
        g_feat = grouping(features, idx)
        return torch.cat([g_xyz, g_feat], dim=1) if self.use_xyz else g_feat


class QueryAndGroup_Edge(nn.Module):
    """pointops.py:519-582.  With knn_dilation > 1 the reference queries
    dilation*nsample candidates and keeps columns randperm(nsample) of them
    (:553-555), i.e. the nsample nearest in a random (CPU-RNG) order."""

    def __init__(self, radius=None, nsample=32, knn_dilation=1, use_xyz=True, ret_gxyz=False, ret_sample_idx=False):
        super().__init__()
        self.radius, self.nsample, self.knn_dilation, self.use_xyz = radius, nsample, knn_dilation, use_xyz
        self.ret_gxyz, self.ret_sample_idx = ret_gxyz, ret_sample_idx

    def forward(self, xyz, new_xyz=None, features=None, center_features=None, idx=None):
        new_xyz = xyz if new_xyz is None else new_xyz
        if idx is None:
            if self.radius is None and self.knn_dilation > 1:
                cand = knnquery(self.knn_dilation * self.nsample, xyz, new_xyz)
                idx = cand[:, :, torch.randperm(self.nsample)].contiguous()
            else:
                idx = _query(self.radius, self.nsample, xyz, new_xyz)
        o_g_xyz = grouping(xyz.transpose(1, 2).contiguous(), idx)
        g_xyz = o_g_xyz - new_xyz.transpose(1, 2).unsqueeze(-1)
        if features is not None:
            g_feat = grouping(features, idx)
            if g_feat.size(3) > 1:
                g_feat = g_feat - center_features.unsqueeze(-1)
            res = torch.cat([g_xyz, g_feat], dim=1) if self.use_xyz else g_feat
        else:
            assert self.use_xyz
            res = g_xyz
        if self.ret_gxyz:
            res = res, o_g_xyz
        if self.ret_sample_idx:
            res = res, idx
        return res


class QueryAndGroup_Edge_Split(nn.Module):
    """pointops.py:584-635"""

    def __init__(self, radius=None, nsample=32, use_xyz=True, ret_gxyz=False):
        super().__init__()
        self.radius, self.nsample, self.use_xyz, self.ret_gxyz = radius, nsample, use_xyz, ret_gxyz

    def forward(self, xyz, new_xyz=None, features=None, center_features=None, idx=None):
        new_xyz = xyz if new_xyz is None else new_xyz
        if idx is None:
            idx = _query(self.radius, self.nsample, xyz, new_xyz)
        o_g_xyz = grouping(xyz.transpose(1, 2).contiguous(), idx)
        g_xyz = o_g_xyz - new_xyz.transpose(1, 2).unsqueeze(-1)
        if features is not None:
            g_feat = grouping(features, idx)
            if g_feat.size(3) > 1:
                g_feat = g_feat - center_features.unsqueeze(-1)
            res = torch.cat([g_xyz, g_feat], dim=1) if self.use_xyz else g_feat
        else:
            assert self.use_xyz
            res = g_xyz
        return (res, o_g_xyz) if self.ret_gxyz else (res, g_xyz)


class GroupAll(nn.Module):
    """pointops.py:637-661"""

    def __init__(self, use_xyz=True):
        super().__init__()
        self.use_xyz = use_xyz

    def forward(self, xyz, new_xyz, features=None):
        g_xyz = xyz.transpose(1, 2).unsqueeze(2)
        if features is None:
            return g_xyz
        g_feat = features.unsqueeze(2)
        return torch.cat([g_xyz, g_feat], dim=1) if self.use_xyz else g_feat
