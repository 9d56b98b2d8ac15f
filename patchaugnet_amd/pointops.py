"""Host-side op layer: the reference's ``libs/pointops/functions/pointops.py`` API on MI355X.

Same public names, argument order, return values and autograd behaviour as the
reference (cited per function, paths under the reference tree), but every op is a
call into libpatchaugnet_hip.so through the C ABI (include/patchaugnet_hip.h) on
the current torch stream.  Differences from the reference, all deliberate:

  * outputs are allocated on the input's device (the reference hard-codes
    ``torch.cuda.*Tensor`` = device 0) and calls run under a device guard;
  * native failures raise RuntimeError (the reference calls exit(-1));
  * ``knnquery`` accepts any nsample (the reference overflows a 200-entry local array).

CPU tensors are rejected: there is no fallback path.
"""
import torch
import torch.nn as nn
from torch.autograd import Function

from . import _lib
from ._arena import zeros
from ._lib import call, check_device, ptr


def _guard(t):
    return torch.cuda.device(t.device)


def _new(like, shape, dtype):
    return torch.empty(shape, dtype=dtype, device=like.device)


class FurthestSampling(Function):
    """pointops.py:11-29 -- xyz (b,n,3) -> idx (b,m) int32."""

    @staticmethod
    def forward(ctx, xyz, m):
        check_device(xyz)
        b, n, _ = xyz.shape
        idx = _new(xyz, (b, m), torch.int32)
        temp = torch.full((b, n), 1e10, dtype=torch.float32, device=xyz.device)
        with _guard(xyz):
            call("pa_furthestsampling", b, n, m, ptr(xyz), ptr(temp), ptr(idx))
        ctx.mark_non_differentiable(idx)
        return idx

    @staticmethod
    def backward(ctx, a=None):
        return None, None


furthestsampling = FurthestSampling.apply


def furthestsampling_gather(xyz, m):
    """(idx (b,m) int32, new_xyz (b,m,3)): furthestsampling plus the gathering of the sampled coordinates (patch_aug_net.py:222-225) as the one launch
    the inference engine uses (pa_furthestsampling_gather: running minima in registers, no temp tensor).  When the coordinates carry a gradient
    (train.run_model(input_grad=True): the reference's feed.requires_grad_) the two differentiable ops run instead -- new_xyz feeds the centred
    neighbour coordinates, so d loss / d xyz has a term through it."""
    check_device(xyz)
    b, n, _ = xyz.shape
    if n > 8192 or xyz.dtype != torch.float32 or (torch.is_grad_enabled() and xyz.requires_grad):
        idx = furthestsampling(xyz, m)
        return idx, gathering(xyz.transpose(1, 2).contiguous(), idx).transpose(1, 2).contiguous()
    with torch.no_grad():
        xyz = xyz.contiguous()
        idx = _new(xyz, (b, m), torch.int32)
        new_xyz = _new(xyz, (b, m, 3), torch.float32)
        with _guard(xyz):
            call("pa_furthestsampling_gather", b, n, m, ptr(xyz), ptr(idx), ptr(new_xyz))
    return idx, new_xyz


@torch.no_grad()
def three_nn_weights(unknown, known):
    """(idx (b,n,3) int32, weight (b,n,3)): nearestneighbor plus the FP module's inverse-distance weights (patch_aug_net.py:350-353: d = sqrt(d2),
    r = 1 / (d + 1e-8), w = r / sum r) in one launch; the search has no gradient (pointops.py:79-82), so neither have the weights."""
    unknown, known = unknown.contiguous(), known.contiguous()
    check_device(unknown, known)
    b, n, _ = unknown.shape
    weight = _new(unknown, (b, n, 3), torch.float32)
    idx = _new(unknown, (b, n, 3), torch.int32)
    with _guard(unknown):
        call("pa_three_nn_weights", b, n, known.shape[1], ptr(unknown), ptr(known), ptr(weight), ptr(idx))
    return idx, weight


@torch.no_grad()
def compose_indices(table, idx):
    """table (b,n) int32 indexed by idx (b,...) int32 along the last axis of table: torch.gather(table[:, None].expand(...), -1, idx.long()) without the
    int64 copy of idx."""
    table, idx = table.contiguous(), idx.contiguous()
    check_device(table, idx)
    out = torch.empty_like(idx)
    with _guard(table):
        call("pa_compose_indices", table.shape[0], table.shape[1], idx[0].numel(), ptr(table), ptr(idx), ptr(out))
    return out


class Gathering(Function):
    """pointops.py:32-57 -- features (b,c,n), idx (b,m) -> (b,c,m)."""

    @staticmethod
    def forward(ctx, features, idx):
        check_device(features, idx)
        b, c, n = features.shape
        m = idx.shape[1]
        out = _new(features, (b, c, m), torch.float32)
        with _guard(features):
            call("pa_gathering_forward", b, c, n, m, ptr(features), ptr(idx), ptr(out))
        ctx.for_backwards = (idx, c, n)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        idx, c, n = ctx.for_backwards
        b, m = idx.shape
        grad = zeros((b, c, n), torch.float32, grad_out.device)
        g = grad_out.contiguous()
        with _guard(g):
            call("pa_gathering_backward", b, c, n, m, ptr(g), ptr(idx), ptr(grad))
        return grad, None


gathering = Gathering.apply


class NearestNeighbor(Function):
    """pointops.py:60-82 -- unknown (b,n,3), known (b,m,3) -> (sqrt(dist2) (b,n,3), idx (b,n,3))."""

    @staticmethod
    def forward(ctx, unknown, known):
        check_device(unknown, known)
        b, n, _ = unknown.shape
        m = known.shape[1]
        dist2 = _new(unknown, (b, n, 3), torch.float32)
        idx = _new(unknown, (b, n, 3), torch.int32)
        with _guard(unknown):
            call("pa_nearestneighbor", b, n, m, ptr(unknown), ptr(known), ptr(dist2), ptr(idx))
        ctx.mark_non_differentiable(idx)
        return torch.sqrt(dist2), idx

    @staticmethod
    def backward(ctx, a=None, b=None):
        return None, None


nearestneighbor = NearestNeighbor.apply


def _gather_form(b, c, n, m):
    """Shapes the atomics-free interpolation backward is built for (csrc/gather.hip)."""
    return 1024 <= n <= 4096 and m <= 8192 and c >= 16


def interpolation_backward_lists(idx, weight, m):
    """The (point, neighbour) lists of interpolation's backward pass inverted per known point -- coordinate-only work a training loop can run
    ahead of the step (backbone geometry(), prefetched with the neighbour searches); pass the result as interpolation's 4th argument."""
    check_device(idx, weight)
    b, n = idx.shape[:2]
    scratch = torch.empty(_lib.lib().pa_interpolation_backward_scratch_ints(b, n, m), dtype=torch.int32, device=idx.device)
    with _guard(idx):
        call("pa_interpolation_backward_lists", b, n, m, ptr(idx), ptr(weight), ptr(scratch))
    return scratch


class Interpolation(Function):
    """pointops.py:85-118 -- features (b,c,m), idx/weight (b,n,3) -> (b,c,n).  lists: optional interpolation_backward_lists(idx, weight, m)."""

    @staticmethod
    def forward(ctx, features, idx, weight, lists=None):
        check_device(features, idx, weight)
        b, c, m = features.shape
        n = idx.shape[1]
        ctx.interpolation_for_backward = (idx, weight, m, lists)
        out = _new(features, (b, c, n), torch.float32)
        with _guard(features):
            call("pa_interpolation_forward", b, c, m, n, ptr(features), ptr(idx), ptr(weight), ptr(out))
        return out

    @staticmethod
    def backward(ctx, grad_out):
        idx, weight, m, lists = ctx.interpolation_for_backward
        b, c, n = grad_out.shape
        grad = zeros((b, c, m), torch.float32, grad_out.device)
        gather = _gather_form(b, c, n, m)
        # a channel slice of a wider contiguous tensor (the backward of the torch.cat that appended the skip features) is read in place
        sliced = gather and grad_out.dtype == torch.float32 and grad_out.stride(2) == 1 and grad_out.stride(1) == n and grad_out.stride(0) >= c * n
        g = grad_out if sliced else grad_out.contiguous()
        gstride = g.stride(0) if sliced else 0
        with _guard(g):
            if gather:      # atomics-free form: one inversion of the index list, then plain sums
                if lists is not None:
                    call("pa_interpolation_backward_gather", b, c, n, m, ptr(g), gstride, None, None, ptr(grad), ptr(lists))
                else:
                    scratch = torch.empty(_lib.lib().pa_interpolation_backward_scratch_ints(b, n, m), dtype=torch.int32, device=g.device)
                    call("pa_interpolation_backward_gather", b, c, n, m, ptr(g), gstride, ptr(idx), ptr(weight), ptr(grad), ptr(scratch))
            else:
                call("pa_interpolation_backward", b, c, n, m, ptr(g), ptr(idx), ptr(weight), ptr(grad))
        return grad, None, None, None


interpolation = Interpolation.apply


class Grouping(Function):
    """pointops.py:121-150 -- features (b,c,n), idx (b,m,nsample) -> (b,c,m,nsample)."""

    @staticmethod
    def forward(ctx, features, idx):
        check_device(features, idx)
        b, c, n = features.shape
        _, m, nsample = idx.shape
        out = _new(features, (b, c, m, nsample), torch.float32)
        with _guard(features):
            call("pa_grouping_forward", b, c, n, m, nsample, ptr(features), ptr(idx), ptr(out))
        ctx.for_backwards = (idx, n)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        idx, n = ctx.for_backwards
        b, c, m, nsample = grad_out.shape
        grad = zeros((b, c, n), torch.float32, grad_out.device)
        g = grad_out.contiguous()
        with _guard(g):
            call("pa_grouping_backward", b, c, n, m, nsample, ptr(g), ptr(idx), ptr(grad))
        return grad, None


grouping = Grouping.apply


class GroupingInt(Function):
    """pointops.py:153-172 -- int64 payload."""

    @staticmethod
    def forward(ctx, features, idx):
        check_device(features, idx)
        b, c, n = features.shape
        _, m, nsample = idx.shape
        out = _new(features, (b, c, m, nsample), torch.int64)
        with _guard(features):
            call("pa_grouping_int_forward", b, c, n, m, nsample, ptr(features), ptr(idx), ptr(out))
        return out

    @staticmethod
    def backward(ctx, a=None):
        return None, None


grouping_int = GroupingInt.apply


class BallQuery(Function):
    """pointops.py:175-197 -- note the native argument order (new_xyz, xyz)."""

    @staticmethod
    def forward(ctx, radius, nsample, xyz, new_xyz):
        check_device(xyz, new_xyz)
        b, n, _ = xyz.shape
        m = new_xyz.shape[1]
        idx = torch.zeros((b, m, nsample), dtype=torch.int32, device=xyz.device)
        with _guard(xyz):
            call("pa_ballquery", b, n, m, float(radius), nsample, ptr(new_xyz), ptr(xyz), ptr(idx))
        ctx.mark_non_differentiable(idx)
        return idx

    @staticmethod
    def backward(ctx, a=None):
        return None, None, None, None


ballquery = BallQuery.apply


class FeatureDistribute(Function):
    """pointops.py:200-221"""

    @staticmethod
    def forward(ctx, max_xyz, xyz):
        check_device(max_xyz, xyz)
        b, n, _ = max_xyz.shape
        m = xyz.shape[1]
        out = torch.zeros((b, m), dtype=torch.int32, device=xyz.device)
        with _guard(xyz):
            call("pa_featuredistribute", b, n, m, ptr(max_xyz), ptr(xyz), ptr(out))
        ctx.mark_non_differentiable(out)
        return out

    @staticmethod
    def backward(ctx, a=None):
        return None, None


featuredistribute = FeatureDistribute.apply


class FeatureGather(Function):
    """pointops.py:224-256"""

    @staticmethod
    def forward(ctx, max_feature, distribute_idx):
        check_device(max_feature, distribute_idx)
        b, c, n = max_feature.shape
        m = distribute_idx.shape[1]
        out = torch.zeros((b, c, m), dtype=torch.float32, device=max_feature.device)
        with _guard(max_feature):
            call("pa_featuregather_forward", b, n, m, c, ptr(max_feature), ptr(distribute_idx), ptr(out))
        ctx.for_backwards = (distribute_idx, n)
        return out

    @staticmethod
    def backward(ctx, grad):
        distribute_idx, n = ctx.for_backwards
        b, c, m = grad.shape
        out = zeros((b, c, n), torch.float32, grad.device)
        g = grad.contiguous()
        with _guard(g):
            call("pa_featuregather_backward", b, n, m, c, ptr(g), ptr(distribute_idx), ptr(out))
        return out, None


featuregather = FeatureGather.apply


class LabelStatBallRange(Function):
    """pointops.py:259-285"""

    @staticmethod
    def forward(ctx, radius, xyz, new_xyz, label_stat):
        check_device(xyz, new_xyz, label_stat)
        b, n, nclass = label_stat.shape
        m = new_xyz.shape[1]
        out = torch.zeros((b, m, nclass), dtype=torch.int32, device=xyz.device)
        with _guard(xyz):
            call("pa_labelstat_ballrange", b, n, m, float(radius), nclass, ptr(new_xyz), ptr(xyz), ptr(label_stat), ptr(out))
        return out

    @staticmethod
    def backward(ctx, a=None):
        return None, None, None, None


labelstat_ballrange = LabelStatBallRange.apply


class LabelStatIdx(Function):
    """pointops.py:288-312"""

    @staticmethod
    def forward(ctx, nsample, label_stat, idx):
        check_device(label_stat, idx)
        b, n, nclass = label_stat.shape
        m = idx.shape[1]
        out = torch.zeros((b, m, nclass), dtype=torch.int32, device=idx.device)
        with _guard(idx):
            call("pa_labelstat_idx", b, n, m, nsample, nclass, ptr(label_stat), ptr(idx), ptr(out))
        return out

    @staticmethod
    def backward(ctx, a=None):
        return None, None, None


labelstat_idx = LabelStatIdx.apply


class LabelStatAndBallQuery(Function):
    """pointops.py:315-344"""

    @staticmethod
    def forward(ctx, radius, nsample, xyz, new_xyz, label_stat):
        check_device(xyz, new_xyz, label_stat)
        b, n, nclass = label_stat.shape
        m = new_xyz.shape[1]
        out = torch.zeros((b, m, nclass), dtype=torch.int32, device=xyz.device)
        idx = torch.zeros((b, m, nsample), dtype=torch.int32, device=xyz.device)
        with _guard(xyz):
            call("pa_labelstat_and_ballquery", b, n, m, float(radius), nsample, nclass, ptr(new_xyz), ptr(xyz),
                 ptr(label_stat), ptr(idx), ptr(out))
        return out, idx

    @staticmethod
    def backward(ctx, a=None, b=None):
        return None, None, None, None, None


labelstat_and_ballquery = LabelStatAndBallQuery.apply


class _PairwiseDistances(Function):
    """||x_i - y_j||^2 clamped at 0 on the MFMA GEMM kernel (pa_tgemm_nn, act = 2: norms and clamp in the epilogue), with its gradient on the same
    kernel: with G' = g where the clamp was inactive (out > 0), dx = 2 (rowsum(G') x - G' y), dy = 2 (colsum(G') y - G'^T x)."""

    @staticmethod
    def forward(ctx, x, y):
        x, y = x.contiguous(), y.contiguous()
        n, d = x.shape
        m = y.shape[0]
        yt = y.t().contiguous()                                              # (d, M): the GEMM's B operand
        xn, yn = (x * x).sum(1), (y * y).sum(1)
        out = torch.empty((n, m), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            call("pa_tgemm_nn", 1, n, m, d, ptr(x), 0, d, 1, ptr(yt), 0, m, 0, ptr(None), ptr(None), ptr(out), 0, m, 0, ptr(xn), ptr(yn), 2, ptr(None), 0)
        ctx.save_for_backward(x, y, out)
        return out

    @staticmethod
    def backward(ctx, g):
        x, y, out = ctx.saved_tensors
        n, d = x.shape
        m = y.shape[0]
        gm = (g * (out > 0)).contiguous()                                    # torch.clamp(min=0): no gradient where the clamp acted
        dx = dy = None
        with torch.cuda.device(x.device):
            if ctx.needs_input_grad[0]:
                t = torch.empty((n, d), dtype=torch.float32, device=x.device)
                call("pa_tgemm_nn", 1, n, d, m, ptr(gm), 0, m, 1, ptr(y), 0, d, 0, ptr(None), ptr(None), ptr(t), 0, d, 0, ptr(None), ptr(None), 0, ptr(None), 0)
                dx = 2.0 * (gm.sum(1, keepdim=True) * x - t)
            if ctx.needs_input_grad[1]:
                t = torch.empty((m, d), dtype=torch.float32, device=x.device)
                call("pa_tgemm_nn", 1, m, d, n, ptr(gm), 0, m, 0, ptr(x), 0, d, 0, ptr(None), ptr(None), ptr(t), 0, d, 0, ptr(None), ptr(None), 0, ptr(None), 0)
                dy = 2.0 * (gm.sum(0).unsqueeze(1) * y - t)
        return dx, dy


def pairwise_distances(x, y=None):
    """pointops.py:347-363 -- ||x_i - y_j||^2 via the expanded form, clamped at 0.  x (N, d), y (M, d) -> (N, M).
    Two fp32 matrices on the MI355X run on the hand-written MFMA GEMM with the norms and the clamp in its epilogue (pa_tgemm_nn, act = 2:
    csrc/train_gemm.hip -- the kernel the retrieval search uses), with or without autograd (round 6: the gradient runs on the same kernel; until
    then a gradient sent the call to torch.mm).  The reference's plain torch statement remains the form for CPU tensors, mixed devices and
    non-fp32 inputs.  y = None: both operands are x (the gradient of either use accumulates into x)."""
    same = y is None
    y = x if same else y
    if x.is_cuda and y.is_cuda and x.device == y.device and x.dtype == torch.float32 and y.dtype == torch.float32 and x.dim() == 2 and y.dim() == 2:
        return _PairwiseDistances.apply(x, y)
    x_norm = (x ** 2).sum(1).view(-1, 1)
    y_norm = (y ** 2).sum(1).view(1, -1)
    return torch.clamp(x_norm + y_norm - 2.0 * torch.mm(x, y.t()), min=0.0)


def knnquery_with_dist(nsample, xyz, new_xyz=None):
    """The native op with both outputs: idx (b,m,nsample) int32 and dist2 (b,m,nsample) fp32."""
    if new_xyz is None:
        new_xyz = xyz
    check_device(xyz, new_xyz)
    b, m, _ = new_xyz.shape
    n = xyz.shape[1]
    idx = _new(xyz, (b, m, nsample), torch.int32)
    dist2 = _new(xyz, (b, m, nsample), torch.float32)
    with _guard(xyz):
        call("pa_knnquery", b, n, m, nsample, ptr(xyz), ptr(new_xyz), ptr(idx), ptr(dist2))
    return idx, dist2


class KNNQuery(Function):
    """pointops.py:407-433 -- returns idx only (the reference discards dist2 too)."""

    @staticmethod
    def forward(ctx, nsample, xyz, new_xyz=None):
        idx, _ = knnquery_with_dist(nsample, xyz, new_xyz)
        ctx.mark_non_differentiable(idx)
        return idx

    @staticmethod
    def backward(ctx, a=None):
        return None, None, None


knnquery = KNNQuery.apply


class KNNQueryNaive(Function):
    """pointops.py:367-404 -- the reference sorts the full (b, m, n) distance matrix with torch.sort (unstable among equal distances)
    and keeps the first nsample columns; the HIP kNN returns exactly that set in (distance, index) order without the matrix."""

    @staticmethod
    def forward(ctx, nsample, xyz, new_xyz=None):
        idx, _ = knnquery_with_dist(nsample, xyz, new_xyz)
        ctx.mark_non_differentiable(idx)
        return idx

    @staticmethod
    def backward(ctx, a=None):
        return None, None, None


knnquery_naive = KNNQueryNaive.apply


class KNNQueryExclude(Function):
    """pointops.py:436-473 -- columns 1 .. nsample of the sorted distance matrix, i.e. the nsample nearest EXCLUDING the closest
    (the point itself when new_xyz is xyz)."""

    @staticmethod
    def forward(ctx, nsample, xyz, new_xyz=None):
        idx, _ = knnquery_with_dist(nsample + 1, xyz, new_xyz)
        idx = idx[:, :, 1:].contiguous()
        ctx.mark_non_differentiable(idx)
        return idx

    @staticmethod
    def backward(ctx, a=None):
        return None, None, None


knnquery_exclude = KNNQueryExclude.apply


def _neighbours(radius, nsample, xyz, new_xyz):
    return ballquery(radius, nsample, xyz, new_xyz) if radius is not None else knnquery(nsample, xyz, new_xyz)


def grouped_coordinates(xyz, new_xyz, idx):
    """(neighbour coordinates, neighbour coordinates minus their centre), (B, 3, m, k) each: the coordinate-only part of the QueryAndGroup_Edge*
    modules (pointops.py:559-562) -- a training loop computes it with the prefetched neighbour searches (backbone geometry())."""
    if xyz.is_cuda and xyz.dtype == torch.float32 and not (torch.is_grad_enabled() and (xyz.requires_grad or new_xyz.requires_grad)):
        return grouped_coordinates_fused(xyz, new_xyz, idx, 1)
    o_grouped_xyz = grouping(xyz.transpose(1, 2).contiguous(), idx)
    return o_grouped_xyz, o_grouped_xyz - new_xyz.transpose(1, 2).unsqueeze(-1)


@torch.no_grad()
def grouped_coordinates_fused(xyz, new_xyz, idx, reps):
    """grouped_coordinates in one launch (pa_group_xyz); reps = 2 returns the centred coordinates twice along the channel axis, (B, 6, m, k): the first
    level's whole grouped input, whose features are the coordinates (cat([grouped_xyz, grouped_features - centre_features]), pointops.py:563-570)."""
    xyz, new_xyz, idx = xyz.contiguous(), new_xyz.contiguous(), idx.contiguous()
    check_device(xyz, new_xyz, idx)
    b, n, _ = xyz.shape
    _, m, k = idx.shape
    o_grouped = _new(xyz, (b, 3, m, k), torch.float32)
    centred = _new(xyz, (b, 3 * reps, m, k), torch.float32)
    with _guard(xyz):
        call("pa_group_xyz", b, n, m, k, reps, ptr(xyz), ptr(new_xyz), ptr(idx), ptr(o_grouped), ptr(centred))
    return o_grouped, centred


class EdgeGroup(Function):
    """cat([grouped_xyz, grouping(features, idx) - gathering(features, center_idx)[..., None]], 1) as ONE op each way (csrc/group_edge.hip): the
    EdgeConv grouping of pointops.py:559-570 for levels that carry features.  features (b,c,n) take the gradient; grouped_xyz (b,3,m,k) is
    coordinate-only (no gradient)."""

    @staticmethod
    def forward(ctx, features, center_idx, idx, grouped_xyz):
        check_device(features, center_idx, idx, grouped_xyz)
        b, c, n = features.shape
        _, m, k = idx.shape
        out = _new(features, (b, 3 + c, m, k), torch.float32)
        with _guard(features):
            call("pa_group_edge_forward", b, c, n, m, k, ptr(features), ptr(center_idx), ptr(idx), ptr(grouped_xyz), ptr(out))
        ctx.for_backwards = (center_idx, idx, n)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        center_idx, idx, n = ctx.for_backwards
        b, c3, m, k = grad_out.shape
        g = grad_out.contiguous()
        grad = _new(g, (b, c3 - 3, n), torch.float32)
        with _guard(g):
            call("pa_group_edge_backward", b, c3 - 3, n, m, k, ptr(g), ptr(center_idx), ptr(idx), ptr(grad))
        return grad, None, None, None


edge_group = EdgeGroup.apply


def _centred_groups(xyz, new_xyz, features, center_features, idx, use_xyz, coords=None, center_idx=None):
    """Shared tail of the QueryAndGroup_Edge* modules (pointops.py:559-570 / :617-630).  coords: grouped_coordinates(xyz, new_xyz, idx) when the
    caller already has it."""
    o_grouped_xyz, grouped_xyz = coords if coords is not None else grouped_coordinates(xyz, new_xyz, idx)
    if (features is not None and center_idx is not None and use_xyz and features.is_cuda and features.dtype == torch.float32 and idx.shape[2] > 1
            and not grouped_xyz.requires_grad and features.shape[2] <= 16384):
        # centres given by index: gather, group, subtract and concatenate in one launch each way
        return edge_group(features.contiguous(), center_idx.contiguous(), idx, grouped_xyz.contiguous()), o_grouped_xyz, grouped_xyz
    if features is not None and center_features is None:
        center_features = gathering(features, center_idx)
    if features is not None:
        grouped = grouping(features, idx)
        if grouped.size(3) > 1:
            grouped = grouped - center_features.unsqueeze(-1)
        new_features = torch.cat([grouped_xyz, grouped], dim=1) if use_xyz else grouped
    else:
        assert use_xyz, "Cannot have not features and not use xyz as a feature!"
        new_features = grouped_xyz
    return new_features, o_grouped_xyz, grouped_xyz


class QueryAndGroup(nn.Module):
    """pointops.py:476-516 -- kNN (radius None) or ball grouping, xyz centred on the query, features as is."""

    def __init__(self, radius=None, nsample=32, use_xyz=True):
        super().__init__()
        self.radius, self.nsample, self.use_xyz = radius, nsample, use_xyz

    def forward(self, xyz, new_xyz=None, features=None, idx=None):
        new_xyz = xyz if new_xyz is None else new_xyz
        if idx is None:
            idx = _neighbours(self.radius, self.nsample, xyz, new_xyz)
        grouped_xyz = grouping(xyz.transpose(1, 2).contiguous(), idx)
        grouped_xyz -= new_xyz.transpose(1, 2).unsqueeze(-1)
        if features is None:
            assert self.use_xyz, "Cannot have not features and not use xyz as a feature!"
            return grouped_xyz
        grouped = grouping(features, idx)
        return torch.cat([grouped_xyz, grouped], dim=1) if self.use_xyz else grouped


class QueryAndGroup_Edge(nn.Module):
    """pointops.py:519-582 -- EdgeConv-style grouping: neighbour minus centre for xyz AND features.

    With knn_dilation > 1 the reference asks for dilation*nsample neighbours and keeps columns
    ``torch.randperm(nsample)`` of them (:553-555), i.e. the nsample NEAREST in a random order drawn
    from the CPU generator.  Only nsample neighbours are therefore searched here (the sorted top-k
    list's prefix is the same) and the same CPU-generator permutation is applied, so ``sample_idx``
    matches the reference element for element under the same ``torch.manual_seed``."""

    def __init__(self, radius=None, nsample=32, knn_dilation=1, use_xyz=True, ret_gxyz=False, ret_sample_idx=False):
        super().__init__()
        self.radius, self.nsample, self.knn_dilation, self.use_xyz = radius, nsample, knn_dilation, use_xyz
        self.ret_gxyz, self.ret_sample_idx = ret_gxyz, ret_sample_idx
        self.perm_buffer = None     # hipGraph capture (train.GraphedTrainer): a device tensor the caller refreshes with torch.randperm per step

    def neighbours(self, xyz, new_xyz):
        """The neighbour lists forward() groups by (search + the reference's column permutation when knn_dilation > 1)."""
        idx = _neighbours(self.radius, self.nsample, xyz, new_xyz)
        if self.radius is None and self.knn_dilation > 1:
            # a host-generated permutation cannot be drawn inside a captured graph: the capturing caller owns a device buffer instead
            perm = self.perm_buffer if self.perm_buffer is not None else torch.randperm(self.nsample).to(idx.device)
            idx = idx.index_select(2, perm).contiguous()
        return idx

    def forward(self, xyz, new_xyz=None, features=None, center_features=None, idx=None, coords=None, center_idx=None):
        """center_idx (b, m): the centres as indices into the cloud (= what center_features was gathered with); given INSTEAD of center_features
        the level's grouping runs as one fused op each way (EdgeGroup)."""
        new_xyz = xyz if new_xyz is None else new_xyz
        if idx is None:
            idx = self.neighbours(xyz, new_xyz)
        new_features, o_grouped_xyz, _ = _centred_groups(xyz, new_xyz, features, center_features, idx, self.use_xyz, coords, center_idx=center_idx)
        res = new_features
        if self.ret_gxyz:
            res = res, o_grouped_xyz
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
            idx = _neighbours(self.radius, self.nsample, xyz, new_xyz)
        new_features, o_grouped_xyz, grouped_xyz = _centred_groups(xyz, new_xyz, features, center_features, idx, self.use_xyz)
        return (new_features, o_grouped_xyz) if self.ret_gxyz else (new_features, grouped_xyz)


class GroupAll(nn.Module):
    """pointops.py:637-661"""

    def __init__(self, use_xyz=True):
        super().__init__()
        self.use_xyz = use_xyz

    def forward(self, xyz, new_xyz, features=None):
        grouped_xyz = xyz.transpose(1, 2).unsqueeze(2)
        if features is None:
            return grouped_xyz
        grouped = features.unsqueeze(2)
        return torch.cat([grouped_xyz, grouped], dim=1) if self.use_xyz else grouped
