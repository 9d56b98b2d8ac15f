"""KNN_CUDA operator contract (``libs/KNN_CUDA/knn_cuda/__init__.py:41-74``, ``csrc/cuda/knn.cpp:23-56``) on the C ABI.

``knn(ref (dim,nr), query (dim,nq), k) -> (dist (k,nq) fp32 L2, ind (k,nq) int64, 0-based)``;
``KNN(k, transpose_mode)(ref (bs,dim,nr)|(bs,nr,dim), query ...) -> (D, I)`` stacked over the batch.
Used here for the descriptor retrieval step (256-D, k = 26).
"""
import torch
import torch.nn as nn

from ._lib import call, check_device, ptr


def knn_raw(ref, query, k):
    """The native op: 1-based int64 indices like the reference's ``_knn.knn``."""
    ref, query = ref.contiguous(), query.contiguous()
    check_device(ref, query)
    assert ref.dtype == torch.float32 and query.dtype == torch.float32, "ref/query must be float32"
    dim, nr = ref.shape
    nq = query.shape[1]
    assert query.shape[0] == dim, "ref and query must have the same dimension"
    dist = torch.empty((k, nq), dtype=torch.float32, device=ref.device)
    ind = torch.empty((k, nq), dtype=torch.int64, device=ref.device)
    with torch.cuda.device(ref.device):
        call("pa_knn_generic", ptr(ref), nr, ptr(query), nq, dim, k, ptr(dist), ptr(ind))
    return dist, ind


def knn(ref, query, k):
    d, i = knn_raw(ref, query, k)
    i -= 1
    return d, i


def _T(t, mode=False):
    return t.transpose(0, 1).contiguous() if mode else t


class KNN(nn.Module):
    def __init__(self, k, transpose_mode=False):
        super().__init__()
        self.k = k
        self._t = transpose_mode

    def forward(self, ref, query):
        assert ref.size(0) == query.size(0), "ref.shape={} != query.shape={}".format(ref.shape, query.shape)
        with torch.no_grad():
            D, I = [], []
            for bi in range(ref.size(0)):
                d, i = knn(_T(ref[bi], self._t).float(), _T(query[bi], self._t).float(), self.k)
                D.append(_T(d, self._t))
                I.append(_T(i, self._t))
            return torch.stack(D, dim=0), torch.stack(I, dim=0)
