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


def knn_mfma_raw(ref, query, k, block=4096):
    """Same contract and the SAME results as knn_raw (bit for bit), for database-sized problems: squared distances of a block of
    queries on the MFMA pipe (pa_tgemm_nn, squared-distance epilogue), per-query radix select of the k-th + an error-bounded candidate
    band, exact re-rank with the direct-sum arithmetic (csrc/knn_mfma.hip).  Queries whose candidate band overflows are rerun through
    the exact kernel (one host synchronisation on the flag vector)."""
    ref, query = ref.contiguous(), query.contiguous()
    check_device(ref, query)
    assert ref.dtype == torch.float32 and query.dtype == torch.float32, "ref/query must be float32"
    dim, nr = ref.shape
    nq = query.shape[1]
    assert query.shape[0] == dim and k <= nr
    dev = ref.device
    ref_rows, q_rows = ref.t().contiguous(), query.t().contiguous()
    rn, qn = (ref_rows * ref_rows).sum(1), (q_rows * q_rows).sum(1)
    rmax = rn.max().reshape(1)
    dist = torch.empty((k, nq), dtype=torch.float32, device=dev)
    ind = torch.empty((k, nq), dtype=torch.int64, device=dev)
    flags = torch.zeros(nq, dtype=torch.int32, device=dev)
    a = torch.empty((min(block, nq), nr), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        for q0 in range(0, nq, block):
            nb = min(block, nq - q0)
            call("pa_tgemm_nn", 1, nb, nr, dim, ptr(q_rows[q0:]), 0, dim, 1, ptr(ref), 0, nr, 0, ptr(None), ptr(None), ptr(a), 0, nr, 0,
                 ptr(qn[q0:]), ptr(rn), 2, ptr(None), 0)
            call("pa_knn_mfma_select", ptr(a), nr, nb, nr, dim, k, ptr(ref_rows), ptr(q_rows[q0:]), ptr(qn[q0:]), ptr(rmax), q0, nq, ptr(dist), ptr(ind),
                 ptr(flags))
    bad = torch.nonzero(flags).flatten()
    if bad.numel():
        d2, i2 = knn_raw(ref, query[:, bad].contiguous(), k)
        dist[:, bad], ind[:, bad] = d2, i2
    return dist, ind


MFMA_MIN_PAIRS = 1 << 22      # (nr x nq) from which the MFMA formulation wins over the wave-per-query kernel


def knn(ref, query, k):
    big = ref.shape[1] * query.shape[1] >= MFMA_MIN_PAIRS and k <= 512 and ref.shape[1] >= 4 * k
    d, i = knn_mfma_raw(ref, query, k) if big else knn_raw(ref, query, k)
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
