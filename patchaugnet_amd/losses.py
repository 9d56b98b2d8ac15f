"""Training losses around the descriptor path, same names and arguments as
``losses/pointnetvlad_loss.py`` of the reference.

Descriptor losses are small dense tensor expressions (torch autograd, written with broadcasting instead of the reference's
``repeat`` copies); the point-set losses run on the HIP Chamfer / EMD kernels (patchaugnet_amd/chamfer_dist.py,
emd_module.py).  Shapes: q_vec (B,1,D), pos_vecs (B,P,D), neg_vecs (B,Nn,D), other_neg (B,1,D).
"""
import torch
import torch.nn.functional as F
from torch.autograd import Function

from .chamfer_dist import ChamferDistanceL1
from .emd_module import emdModule


class _TupleLossFused(Function):
    """triplet_loss / quadruplet_loss on the MI355X as ONE launch (csrc/losses.hip: value + gradient with respect to every descriptor); the
    torch statements below are ~25 small kernels forward and ~40 backward on a few KB."""

    @staticmethod
    def forward(ctx, desc, P, Nn, m1, m2, use_min, lazy, ignore_zero, quad):
        from ._lib import call, ptr
        B, T, D = desc.shape
        loss = torch.empty(1, dtype=torch.float32, device=desc.device)
        grad = torch.empty_like(desc)
        with torch.cuda.device(desc.device):
            call("pa_quadruplet_loss", B, P, Nn, D, ptr(desc), float(m1), float(m2), int(use_min), int(lazy), int(ignore_zero), int(quad), ptr(loss), ptr(grad))
        ctx.save_for_backward(grad)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return grad * g, None, None, None, None, None, None, None, None


def _fused_tuple_loss(q_vec, pos_vecs, neg_vecs, other_neg, m1, m2, use_min, lazy, ignore_zero_loss, quad):
    """None when the fused kernel does not apply (CPU tensors, more than 64 tuples / positives / negatives)."""
    if not (q_vec.is_cuda and q_vec.dtype == torch.float32 and q_vec.shape[0] <= 64 and pos_vecs.shape[1] <= 64 and neg_vecs.shape[1] <= 64):
        return None
    parts = [q_vec, pos_vecs, neg_vecs, other_neg if other_neg is not None else q_vec]
    return _TupleLossFused.apply(torch.cat(parts, 1).contiguous(), pos_vecs.shape[1], neg_vecs.shape[1], m1, m2, use_min, lazy, ignore_zero_loss, quad)


def best_pos_distance(query, pos_vecs):
    """pointnetvlad_loss.py:9-15 -- min / max over the positives of the squared descriptor distance."""
    diff = ((pos_vecs - query) ** 2).sum(2)
    return diff.min(1)[0], diff.max(1)[0]


def _reduce(loss, lazy, ignore_zero_loss, lazy_false_mean):
    """Shared tail of triplet / quadruplet terms (:34-45, :76-86): max (lazy) or sum/mean over negatives, then batch mean
    or mean over the non-zero ("hard") rows."""
    loss = loss.max(1)[0] if lazy else (loss.mean(1) if lazy_false_mean else loss.sum(1))
    if ignore_zero_loss:
        hard = torch.gt(loss, 1e-16).float().sum()
        return loss.sum() / (hard + 1e-16)
    return loss.mean()


def triplet_loss(q_vec, pos_vecs, neg_vecs, margin, use_min=False, lazy=False, ignore_zero_loss=False):
    """pointnetvlad_loss.py:18-45"""
    fused = _fused_tuple_loss(q_vec, pos_vecs, neg_vecs, None, margin, 0.0, use_min, lazy, ignore_zero_loss, quad=False)
    if fused is not None:
        return fused
    min_pos, max_pos = best_pos_distance(q_vec, pos_vecs)
    positive = (min_pos if use_min else max_pos).view(-1, 1)
    loss = (margin + positive - ((neg_vecs - q_vec) ** 2).sum(2)).clamp(min=0.0)
    return _reduce(loss, lazy, ignore_zero_loss, lazy_false_mean=False)


def triplet_loss_wrapper(q_vec, pos_vecs, neg_vecs, other_neg, m1, m2, use_min=False, lazy=False, ignore_zero_loss=False):
    """pointnetvlad_loss.py:48-50"""
    return triplet_loss(q_vec, pos_vecs, neg_vecs, m1, use_min, lazy, ignore_zero_loss)


def _hinge(x, soft_margin):
    return torch.log(1 + torch.exp(x.clamp(max=88))) if soft_margin else x.clamp(min=0.0)


def quadruplet_loss(q_vec, pos_vecs, neg_vecs, other_neg, m1, m2, use_min=False, lazy=False, ignore_zero_loss=False, soft_margin=False):
    """pointnetvlad_loss.py:53-105 -- the training loss of configs/patch_aug_net.yaml (LOSS_FUNCTION quadruplet)."""
    if not soft_margin:
        fused = _fused_tuple_loss(q_vec, pos_vecs, neg_vecs, other_neg, m1, m2, use_min, lazy, ignore_zero_loss, quad=True)
        if fused is not None:
            return fused
    min_pos, max_pos = best_pos_distance(q_vec, pos_vecs)
    positive = (min_pos if use_min else max_pos).view(-1, 1)
    first = _hinge(m1 + positive - ((neg_vecs - q_vec) ** 2).sum(2), soft_margin)
    second = _hinge(m2 + positive - ((neg_vecs - other_neg) ** 2).sum(2), soft_margin)
    return _reduce(first, lazy, ignore_zero_loss, True) + _reduce(second, lazy, ignore_zero_loss, True)


def contrastive_quadruplet_loss(q_vec, pos_vecs, neg_vecs, other_neg, m1, m2, use_min=False, lazy=True, ignore_zero_loss=False,
                                soft_margin=False):
    """pointnetvlad_loss.py:108-152 -- hardest-negative triplet term (positive detached where the negative is closer) + second term."""
    min_pos, max_pos = best_pos_distance(q_vec, pos_vecs)
    positive = min_pos if use_min else max_pos
    batch = q_vec.shape[0]
    min_neg = ((neg_vecs - q_vec) ** 2).sum(2).min(1)[0]
    mask = min_neg < positive
    loss1 = loss2 = 0
    if mask.sum() != 0:
        loss1 = (m1 + positive[mask].detach() - min_neg[mask]).clamp(min=0.0).sum()
    if (~mask).sum() != 0:
        loss2 = (m1 + positive[~mask] - min_neg[~mask]).clamp(min=0.0).sum()
    first = (loss1 + loss2) / batch
    second = (m2 + positive.view(-1, 1) - ((neg_vecs - other_neg) ** 2).sum(2)).clamp(min=0.0)
    return first + _reduce(second, lazy, ignore_zero_loss, True)


def hphn_quadruplet_loss(q_vec, pos_vecs, neg_vecs, other_neg, m1, m2, use_min=False, lazy=False, ignore_zero_loss=False):
    """pointnetvlad_loss.py:155-166 -- hardest positive, hardest negative."""
    _, max_pos = best_pos_distance(q_vec, pos_vecs)
    min_neg, _ = best_pos_distance(q_vec, neg_vecs)
    min_other_neg, _ = best_pos_distance(other_neg, neg_vecs)
    return (m1 + max_pos - torch.minimum(min_neg, min_other_neg)).clamp(min=0.0).mean()


def contrastive_loss(q_vec, pos_vec, neg_vec, margin):
    """pointnetvlad_loss.py:169-186 -- lists of (D,) patch features; pairwise_distance adds its eps = 1e-6 to the difference."""
    total = 0.0
    q = torch.stack(q_vec, dim=0)
    if len(pos_vec) > 0:
        total = total + torch.mean(torch.pow(F.pairwise_distance(q, torch.stack(pos_vec, dim=0)), 2))
    if len(neg_vec) > 0:
        total = total + torch.mean(torch.pow(torch.clamp(margin - F.pairwise_distance(q, torch.stack(neg_vec, dim=0)), min=0.0), 2))
    return total


def _cat_float(pcs):
    return torch.cat([p.float() for p in pcs], 1)


def chamfer_loss(pc1, pc2):
    """pointnetvlad_loss.py:189-202"""
    return ChamferDistanceL1()(_cat_float(pc1).squeeze(0), _cat_float(pc2).squeeze(0))


def emd_loss(pc1, pc2):
    """pointnetvlad_loss.py:205-221 -- whole clouds (-1, 4096, 3), eps 0.02, 1024 auction rounds."""
    dis, _ = emdModule()(_cat_float(pc1).view((-1, 4096, 3)), _cat_float(pc2).view((-1, 4096, 3)), 0.02, 1024)
    return torch.mean(torch.mean(torch.sqrt(dis), dim=1))


def point_pair_loss(pc1, pc2):
    """pointnetvlad_loss.py:224-239"""
    return torch.mean(torch.nn.PairwiseDistance(p=2)(_cat_float(pc1).view((-1, 4096, 3)), _cat_float(pc2).view((-1, 4096, 3))))


def patch_chamfer_loss(origin_patches, recon_patches):
    """pointnetvlad_loss.py:242-247 -- L1 Chamfer between (R*1024, 20, 3) patch sets."""
    return ChamferDistanceL1()(torch.cat(origin_patches, 0), torch.cat(recon_patches, 0))


def patch_emd_loss(origin_patches, recon_patches):
    """pointnetvlad_loss.py:250-256.  With 20-point patches the native side rejects the shape (n % 1024 != 0) and the reference
    silently trains on zeros; here that case raises instead."""
    feed, res = torch.cat(origin_patches, 0), torch.cat(recon_patches, 0)
    if feed.shape[1] % 1024 != 0:
        raise ValueError("patch_emd_loss: the auction EMD needs n %% 1024 == 0 points per set, got %d" % feed.shape[1])
    dis, _ = emdModule()(feed, res, 0.02, 1024)
    return torch.mean(torch.mean(torch.sqrt(dis), dim=1))


def get_loss_func(name):
    """train_place_recognition.py:99-117 naming; like there, any other name selects the triplet wrapper."""
    table = {"triplet": triplet_loss_wrapper, "quadruplet": quadruplet_loss, "contrastive_quadruplet": contrastive_quadruplet_loss,
             "hphn_quadruplet": hphn_quadruplet_loss, "contrastive": contrastive_loss, "chamfer": chamfer_loss, "emd": emd_loss,
             "point_pair": point_pair_loss, "patch_chamfer": patch_chamfer_loss, "patch_emd": patch_emd_loss}
    return table.get(name, triplet_loss_wrapper)
