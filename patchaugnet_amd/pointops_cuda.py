"""The reference's native module name ``pointops_cuda`` on top of the C ABI.

``libs/pointops/src/pointops_api.cpp:15-40`` exports 19 void functions that take sizes first and PRE-ALLOCATED torch
tensors after them; the reference's own ``libs/pointops/functions/pointops.py`` does ``import pointops_cuda`` and calls
them (lines 22, 44, 54, 75, 102, 115, 133, 147, 165, 190, 214, 238, 253, 277, 304, 336, 426).  Putting this file on the
import path under that name makes that unmodified Python layer run on the MI355X:

    import sys, patchaugnet_amd.pointops_cuda as m; sys.modules["pointops_cuda"] = m

Same names, same argument order (note ballquery takes (new_xyz, xyz) but knnquery takes (xyz, new_xyz)).  Launches go to
the current torch stream; errors raise RuntimeError instead of exit(-1).
"""
from ._lib import call, check_device, ptr


def _go(name, ints, tensors):
    check_device(*tensors)
    call(name, *ints, *[ptr(t) for t in tensors])


def ballquery_cuda(b, n, m, radius, nsample, new_xyz, xyz, idx):
    check_device(new_xyz, xyz, idx)
    call("pa_ballquery", b, n, m, float(radius), nsample, ptr(new_xyz), ptr(xyz), ptr(idx))


def knnquery_cuda(b, n, m, nsample, xyz, new_xyz, idx, dist2):
    _go("pa_knnquery", (b, n, m, nsample), (xyz, new_xyz, idx, dist2))


def grouping_forward_cuda(b, c, n, m, nsample, points, idx, out):
    _go("pa_grouping_forward", (b, c, n, m, nsample), (points, idx, out))


def grouping_backward_cuda(b, c, n, m, nsample, grad_out, idx, grad_points):
    _go("pa_grouping_backward", (b, c, n, m, nsample), (grad_out, idx, grad_points))


def grouping_int_forward_cuda(b, c, n, m, nsample, points, idx, out):
    _go("pa_grouping_int_forward", (b, c, n, m, nsample), (points, idx, out))


def gathering_forward_cuda(b, c, n, m, points, idx, out):
    _go("pa_gathering_forward", (b, c, n, m), (points, idx, out))


def gathering_backward_cuda(b, c, n, m, grad_out, idx, grad_points):
    _go("pa_gathering_backward", (b, c, n, m), (grad_out, idx, grad_points))


def furthestsampling_cuda(b, n, m, xyz, temp, idx):
    _go("pa_furthestsampling", (b, n, m), (xyz, temp, idx))


def nearestneighbor_cuda(b, n, m, unknown, known, dist2, idx):
    _go("pa_nearestneighbor", (b, n, m), (unknown, known, dist2, idx))


def interpolation_forward_cuda(b, c, m, n, points, idx, weight, out):
    _go("pa_interpolation_forward", (b, c, m, n), (points, idx, weight, out))


def interpolation_backward_cuda(b, c, n, m, grad_out, idx, weight, grad_points):
    _go("pa_interpolation_backward", (b, c, n, m), (grad_out, idx, weight, grad_points))


def labelstat_idx_cuda(b, n, m, nsample, nclass, label_stat, idx, new_label_stat):
    _go("pa_labelstat_idx", (b, n, m, nsample, nclass), (label_stat, idx, new_label_stat))


def labelstat_ballrange_cuda(b, n, m, radius, nclass, new_xyz, xyz, label_stat, new_label_stat):
    check_device(new_xyz, xyz, label_stat, new_label_stat)
    call("pa_labelstat_ballrange", b, n, m, float(radius), nclass, ptr(new_xyz), ptr(xyz), ptr(label_stat), ptr(new_label_stat))


def labelstat_and_ballquery_cuda(b, n, m, radius, nsample, nclass, new_xyz, xyz, label_stat, idx, new_label_stat):
    check_device(new_xyz, xyz, label_stat, idx, new_label_stat)
    call("pa_labelstat_and_ballquery", b, n, m, float(radius), nsample, nclass, ptr(new_xyz), ptr(xyz), ptr(label_stat),
         ptr(idx), ptr(new_label_stat))


def featuredistribute_cuda(b, n, m, max_xyz, xyz, distribute_idx):
    _go("pa_featuredistribute", (b, n, m), (max_xyz, xyz, distribute_idx))


def featuregather_forward_cuda(b, n, m, c, max_feature, distribute_idx, distribute_feature):
    _go("pa_featuregather_forward", (b, n, m, c), (max_feature, distribute_idx, distribute_feature))


def featuregather_backward_cuda(b, n, m, c, grad_distribute_feature, distribute_idx, grad_max_feature):
    _go("pa_featuregather_backward", (b, n, m, c), (grad_distribute_feature, distribute_idx, grad_max_feature))
