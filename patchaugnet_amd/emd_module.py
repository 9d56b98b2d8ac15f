"""Earth mover's distance: the reference's ``emd`` native module and ``libs/emd_module/emd_module.py`` on the C ABI.

``forward(xyz1, xyz2, dist, assignment, price, assignment_inv, bid, bid_increments, max_increments, unass_idx, unass_cnt,
unass_cnt_sum, cnt_tmp, max_idx, eps, iters) -> int`` and ``backward(xyz1, xyz2, gradxyz, graddist, idx) -> int`` keep the
signatures and return codes of ``libs/emd_module/emd.cpp:6-30`` (1 = ok, 0 = launch error, -1 = rejected shape; the
reference prints its reason and so does this).  ``emdFunction`` / ``emdModule`` follow ``emd_module.py:29-78``; tensors are
allocated on the inputs' device instead of the hard-coded ``'cuda'``.
"""
import torch
from torch import nn
from torch.autograd import Function

from . import _lib
from ._lib import check_device, ptr, stream_ptr


def forward(xyz1, xyz2, dist, assignment, price, assignment_inv, bid, bid_increments, max_increments, unass_idx, unass_cnt,
            unass_cnt_sum, cnt_tmp, max_idx, eps, iters):
    """emd_cuda_forward (emd_cuda.cu:228-282).  The four unass_* / cnt_tmp scratch tensors are accepted and left untouched:
    the HIP kernel keeps the list of unassigned points in LDS."""
    check_device(xyz1, xyz2, dist, assignment, price, assignment_inv, bid, bid_increments, max_increments, max_idx)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    l = _lib.lib()
    with torch.cuda.device(xyz1.device):
        rc = l.pa_emd_forward(b, n, m, ptr(xyz1), ptr(xyz2), ptr(dist), ptr(assignment), ptr(price), ptr(assignment_inv), ptr(bid),
                              ptr(bid_increments), ptr(max_increments), ptr(max_idx), float(eps), int(iters), stream_ptr())
    if rc == 0:
        return 1
    msg = l.pa_last_error().decode()
    if rc == -2:                                  # the reference's "Input Error!" cases (emd_cuda.cu:236-249)
        print("Input Error! " + msg)
        return -1
    if rc < 0:
        raise RuntimeError(msg)
    print("error in emd forward: " + msg)         # failed launch: the reference prints and returns 0 (:275-279)
    return 0


def backward(xyz1, xyz2, gradxyz, graddist, idx):
    """emd_cuda_backward (emd_cuda.cu:302-317): gradxyz (zero-filled by the caller) += 2 graddist (xyz1 - xyz2[idx])."""
    check_device(xyz1, xyz2, gradxyz, graddist, idx)
    b, n, _ = xyz1.shape
    with torch.cuda.device(xyz1.device):
        _lib.call("pa_emd_backward", b, n, ptr(xyz1), ptr(xyz2), ptr(gradxyz), ptr(graddist), ptr(idx))
    return 1


_native_forward, _native_backward = forward, backward      # the class below re-uses the names for autograd


class emdFunction(Function):
    @staticmethod
    def forward(ctx, xyz1, xyz2, eps, iters):
        batchsize, n, _ = xyz1.size()
        _, m, _ = xyz2.size()
        assert n == m
        assert xyz1.size(0) == xyz2.size(0)
        assert batchsize <= 512
        xyz1, xyz2 = xyz1.contiguous().float(), xyz2.contiguous().float()
        dev = xyz1.device
        dist = torch.zeros(batchsize, n, device=dev)
        assignment = torch.full((batchsize, n), -1, device=dev, dtype=torch.int32)
        assignment_inv = torch.full((batchsize, m), -1, device=dev, dtype=torch.int32)
        price = torch.zeros(batchsize, m, device=dev)
        bid = torch.zeros(batchsize, n, device=dev, dtype=torch.int32)
        bid_increments = torch.zeros(batchsize, n, device=dev)
        max_increments = torch.zeros(batchsize, m, device=dev)
        max_idx = torch.zeros(batchsize * m, device=dev, dtype=torch.int32)
        _native_forward(xyz1, xyz2, dist, assignment, price, assignment_inv, bid, bid_increments, max_increments, None, None, None,
                        None, max_idx, eps, iters)
        ctx.save_for_backward(xyz1, xyz2, assignment)
        return dist, assignment

    @staticmethod
    def backward(ctx, graddist, gradidx):
        xyz1, xyz2, assignment = ctx.saved_tensors
        graddist = graddist.contiguous()
        gradxyz1 = torch.zeros_like(xyz1)
        gradxyz2 = torch.zeros_like(xyz2)
        _native_backward(xyz1, xyz2, gradxyz1, graddist, assignment)
        return gradxyz1, gradxyz2, None, None


class emdModule(nn.Module):
    def forward(self, input1, input2, eps, iters):
        return emdFunction.apply(input1, input2, eps, iters)
