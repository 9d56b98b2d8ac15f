"""Chamfer distance: the reference's ``chamfer`` native module and ``libs/chamfer_dist/__init__.py`` on the C ABI.

``forward(xyz1 (B,n,3), xyz2 (B,m,3)) -> [dist1, dist2, idx1, idx2]`` and ``backward(...) -> [grad_xyz1, grad_xyz2]`` follow
``libs/chamfer_dist/chamfer_cuda.cpp:12-39`` (they allocate their own outputs); ``ChamferFunction`` / ``ChamferDistanceL1`` /
``L2`` / ``L2_split`` follow ``libs/chamfer_dist/__init__.py:13-84``.
"""
import torch

from ._lib import call, check_device, ptr


def forward(xyz1, xyz2):
    check_device(xyz1, xyz2)
    B, n, _ = xyz1.shape
    m = xyz2.shape[1]
    dist1 = torch.empty((B, n), dtype=torch.float32, device=xyz1.device)
    dist2 = torch.empty((B, m), dtype=torch.float32, device=xyz1.device)
    idx1 = torch.empty((B, n), dtype=torch.int32, device=xyz1.device)
    idx2 = torch.empty((B, m), dtype=torch.int32, device=xyz1.device)
    with torch.cuda.device(xyz1.device):
        call("pa_chamfer_forward", B, n, m, ptr(xyz1), ptr(xyz2), ptr(dist1), ptr(dist2), ptr(idx1), ptr(idx2))
    return [dist1, dist2, idx1, idx2]


def backward(xyz1, xyz2, idx1, idx2, grad_dist1, grad_dist2):
    g1, g2 = grad_dist1.contiguous(), grad_dist2.contiguous()
    check_device(xyz1, xyz2, idx1, idx2, g1, g2)
    B, n, _ = xyz1.shape
    m = xyz2.shape[1]
    grad_xyz1 = torch.empty_like(xyz1)
    grad_xyz2 = torch.empty_like(xyz2)
    with torch.cuda.device(xyz1.device):
        call("pa_chamfer_backward", B, n, m, ptr(xyz1), ptr(xyz2), ptr(idx1), ptr(idx2), ptr(g1), ptr(g2), ptr(grad_xyz1), ptr(grad_xyz2))
    return [grad_xyz1, grad_xyz2]


class ChamferFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xyz1, xyz2):
        xyz1, xyz2 = xyz1.contiguous(), xyz2.contiguous()
        dist1, dist2, idx1, idx2 = forward(xyz1, xyz2)
        ctx.save_for_backward(xyz1, xyz2, idx1, idx2)
        return dist1, dist2

    @staticmethod
    def backward(ctx, grad_dist1, grad_dist2):
        xyz1, xyz2, idx1, idx2 = ctx.saved_tensors
        return tuple(backward(xyz1, xyz2, idx1, idx2, grad_dist1, grad_dist2))


def _strip_zero_points(xyz1, xyz2, ignore_zeros):
    """__init__.py:36-41 -- with batch 1 and ignore_zeros, drop points whose coordinates sum to 0."""
    if xyz1.size(0) == 1 and ignore_zeros:
        xyz1 = xyz1[torch.sum(xyz1, dim=2).ne(0)].unsqueeze(dim=0)
        xyz2 = xyz2[torch.sum(xyz2, dim=2).ne(0)].unsqueeze(dim=0)
    return xyz1, xyz2


class ChamferDistanceL2(torch.nn.Module):
    def __init__(self, ignore_zeros=False):
        super().__init__()
        self.ignore_zeros = ignore_zeros

    def forward(self, xyz1, xyz2):
        d1, d2 = ChamferFunction.apply(*_strip_zero_points(xyz1, xyz2, self.ignore_zeros))
        return torch.mean(d1) + torch.mean(d2)


class ChamferDistanceL2_split(ChamferDistanceL2):
    def forward(self, xyz1, xyz2):
        d1, d2 = ChamferFunction.apply(*_strip_zero_points(xyz1, xyz2, self.ignore_zeros))
        return torch.mean(d1), torch.mean(d2)


class ChamferL1Function(torch.autograd.Function):
    """``(mean(sqrt(dist1)) + mean(sqrt(dist2))) / 2`` (__init__.py:79-84) with the square roots, means and halving inside the
    native calls: one value kernel after the two searches, and the chain rule applied per point inside the gradient scatter."""

    @staticmethod
    def forward(ctx, xyz1, xyz2):
        xyz1, xyz2 = xyz1.contiguous(), xyz2.contiguous()
        check_device(xyz1, xyz2)
        B, n, _ = xyz1.shape
        m = xyz2.shape[1]
        dist = torch.empty((B * (n + m),), dtype=torch.float32, device=xyz1.device)
        idx = torch.empty((B * (n + m),), dtype=torch.int32, device=xyz1.device)
        loss = torch.empty((), dtype=torch.float32, device=xyz1.device)
        partial = torch.empty((B * (-(-n // 256) - (-m // 256)),), dtype=torch.float64, device=xyz1.device)
        with torch.cuda.device(xyz1.device):
            call("pa_chamfer_l1_forward", B, n, m, ptr(xyz1), ptr(xyz2), ptr(dist), ptr(dist[B * n:]), ptr(idx), ptr(idx[B * n:]), ptr(loss), ptr(partial))
        ctx.save_for_backward(xyz1, xyz2, dist, idx)
        return loss

    @staticmethod
    def backward(ctx, gout):
        xyz1, xyz2, dist, idx = ctx.saved_tensors
        B, n, _ = xyz1.shape
        m = xyz2.shape[1]
        gout = gout.contiguous().float()
        grad_xyz1 = torch.empty_like(xyz1)
        grad_xyz2 = torch.empty_like(xyz2)
        with torch.cuda.device(xyz1.device):
            call("pa_chamfer_l1_backward", B, n, m, ptr(xyz1), ptr(xyz2), ptr(idx), ptr(idx[B * n:]), ptr(dist), ptr(dist[B * n:]), ptr(gout),
                 ptr(grad_xyz1), ptr(grad_xyz2))
        return grad_xyz1, grad_xyz2


class ChamferDistanceL1(ChamferDistanceL2):
    def forward(self, xyz1, xyz2):
        return ChamferL1Function.apply(*_strip_zero_points(xyz1, xyz2, self.ignore_zeros))


def patch_chamfer_loss(origin_patches, recon_patches):
    """losses/pointnetvlad_loss.py:242-247: L1 Chamfer between lists of (1024, 20, 3) patch sets."""
    return ChamferDistanceL1()(torch.cat(origin_patches, 0), torch.cat(recon_patches, 0))
