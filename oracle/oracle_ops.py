"""ctypes front-end of the CPU oracle (oracle/pointops_oracle.c).

TEST INFRASTRUCTURE ONLY.  Importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg; never from patchaugnet_amd/.  All functions take and
return numpy arrays (C-contiguous fp32 / int32) and follow the argument order of
the reference's pybind module ``pointops_cuda`` (libs/pointops/src/pointops_api.cpp:15-40).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle_pointops.so")
_lib = None

c_int, c_float = ctypes.c_int, ctypes.c_float
_fp = ctypes.POINTER(ctypes.c_float)
_ip = ctypes.POINTER(ctypes.c_int)
_lp = ctypes.POINTER(ctypes.c_int64)


def build(force=False):
    """Compile the C oracle with gcc (oracle/Makefile)."""
    src = os.path.join(_HERE, "pointops_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "liboracle_pointops.so"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
        _lib.oracle_opt_n_threads.restype = c_int
        _lib.oracle_emd_forward.restype = c_int
    return _lib


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(_fp)


def _i(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(_ip)


def opt_n_threads(n):
    return int(lib().oracle_opt_n_threads(int(n)))


def furthestsampling(xyz, m):
    """xyz (b,n,3) -> idx (b,m) int32.  temp is filled with 1e10 like pointops.py:21."""
    xyz, px = _f(xyz)
    b, n, _ = xyz.shape
    temp = np.full((b, n), 1e10, dtype=np.float32)
    idx = np.zeros((b, m), dtype=np.int32)
    lib().oracle_furthestsampling(b, n, m, px, temp.ctypes.data_as(_fp), idx.ctypes.data_as(_ip))
    return idx


def gathering_forward(points, idx):
    points, pp = _f(points)
    idx, pi = _i(idx)
    b, c, n = points.shape
    m = idx.shape[1]
    out = np.empty((b, c, m), dtype=np.float32)
    lib().oracle_gathering_forward(b, c, n, m, pp, pi, out.ctypes.data_as(_fp))
    return out


def gathering_backward(grad_out, idx, n):
    grad_out, pg = _f(grad_out)
    idx, pi = _i(idx)
    b, c, m = grad_out.shape
    out = np.zeros((b, c, n), dtype=np.float32)
    lib().oracle_gathering_backward(b, c, n, m, pg, pi, out.ctypes.data_as(_fp))
    return out


def knnquery(nsample, xyz, new_xyz):
    """-> (idx (b,m,nsample) int32, dist2 (b,m,nsample) fp32)"""
    xyz, px = _f(xyz)
    new_xyz, pq = _f(new_xyz)
    b, n, _ = xyz.shape
    m = new_xyz.shape[1]
    idx = np.zeros((b, m, nsample), dtype=np.int32)
    d2 = np.zeros((b, m, nsample), dtype=np.float32)
    lib().oracle_knnquery(b, n, m, nsample, px, pq, idx.ctypes.data_as(_ip), d2.ctypes.data_as(_fp))
    return idx, d2


def grouping_forward(points, idx):
    points, pp = _f(points)
    idx, pi = _i(idx)
    b, c, n = points.shape
    _, m, ns = idx.shape
    out = np.empty((b, c, m, ns), dtype=np.float32)
    lib().oracle_grouping_forward(b, c, n, m, ns, pp, pi, out.ctypes.data_as(_fp))
    return out


def grouping_backward(grad_out, idx, n):
    grad_out, pg = _f(grad_out)
    idx, pi = _i(idx)
    b, c, m, ns = grad_out.shape
    out = np.zeros((b, c, n), dtype=np.float32)
    lib().oracle_grouping_backward(b, c, n, m, ns, pg, pi, out.ctypes.data_as(_fp))
    return out


def grouping_int_forward(points, idx):
    points = np.ascontiguousarray(points, dtype=np.int64)
    idx, pi = _i(idx)
    b, c, n = points.shape
    _, m, ns = idx.shape
    out = np.empty((b, c, m, ns), dtype=np.int64)
    lib().oracle_grouping_int_forward(b, c, n, m, ns, points.ctypes.data_as(_lp), pi, out.ctypes.data_as(_lp))
    return out


def nearestneighbor(unknown, known):
    """-> (dist2 (b,n,3) fp32 SQUARED, idx (b,n,3) int32); the sqrt is applied by the Python op layer."""
    unknown, pu = _f(unknown)
    known, pk = _f(known)
    b, n, _ = unknown.shape
    m = known.shape[1]
    d2 = np.empty((b, n, 3), dtype=np.float32)
    idx = np.empty((b, n, 3), dtype=np.int32)
    lib().oracle_nearestneighbor(b, n, m, pu, pk, d2.ctypes.data_as(_fp), idx.ctypes.data_as(_ip))
    return d2, idx


def interpolation_forward(points, idx, weight):
    points, pp = _f(points)
    idx, pi = _i(idx)
    weight, pw = _f(weight)
    b, c, m = points.shape
    n = idx.shape[1]
    out = np.empty((b, c, n), dtype=np.float32)
    lib().oracle_interpolation_forward(b, c, m, n, pp, pi, pw, out.ctypes.data_as(_fp))
    return out


def interpolation_backward(grad_out, idx, weight, m):
    grad_out, pg = _f(grad_out)
    idx, pi = _i(idx)
    weight, pw = _f(weight)
    b, c, n = grad_out.shape
    out = np.zeros((b, c, m), dtype=np.float32)
    lib().oracle_interpolation_backward(b, c, n, m, pg, pi, pw, out.ctypes.data_as(_fp))
    return out


def ballquery(radius, nsample, xyz, new_xyz):
    xyz, px = _f(xyz)
    new_xyz, pq = _f(new_xyz)
    b, n, _ = xyz.shape
    m = new_xyz.shape[1]
    idx = np.zeros((b, m, nsample), dtype=np.int32)
    lib().oracle_ballquery(b, n, m, c_float(radius), nsample, pq, px, idx.ctypes.data_as(_ip))
    return idx


def featuredistribute(max_xyz, xyz):
    max_xyz, pm = _f(max_xyz)
    xyz, px = _f(xyz)
    b, n, _ = max_xyz.shape
    m = xyz.shape[1]
    out = np.zeros((b, m), dtype=np.int32)
    lib().oracle_featuredistribute(b, n, m, pm, px, out.ctypes.data_as(_ip))
    return out


def labelstat_idx(label_stat, idx):
    label_stat, pl = _i(label_stat)
    idx, pi = _i(idx)
    b, n, nclass = label_stat.shape
    _, m, ns = idx.shape
    out = np.zeros((b, m, nclass), dtype=np.int32)
    lib().oracle_labelstat_idx(b, n, m, ns, nclass, pl, pi, out.ctypes.data_as(_ip))
    return out


def labelstat_ballrange(radius, xyz, new_xyz, label_stat):
    xyz, px = _f(xyz)
    new_xyz, pq = _f(new_xyz)
    label_stat, pl = _i(label_stat)
    b, n, nclass = label_stat.shape
    m = new_xyz.shape[1]
    out = np.zeros((b, m, nclass), dtype=np.int32)
    lib().oracle_labelstat_ballrange(b, n, m, c_float(radius), nclass, pq, px, pl, out.ctypes.data_as(_ip))
    return out


def labelstat_and_ballquery(radius, nsample, xyz, new_xyz, label_stat):
    xyz, px = _f(xyz)
    new_xyz, pq = _f(new_xyz)
    label_stat, pl = _i(label_stat)
    b, n, nclass = label_stat.shape
    m = new_xyz.shape[1]
    out = np.zeros((b, m, nclass), dtype=np.int32)
    idx = np.zeros((b, m, nsample), dtype=np.int32)
    lib().oracle_labelstat_and_ballquery(b, n, m, c_float(radius), nsample, nclass, pq, px, pl,
                                         idx.ctypes.data_as(_ip), out.ctypes.data_as(_ip))
    return out, idx


def chamfer_forward(xyz1, xyz2):
    xyz1, p1 = _f(xyz1)
    xyz2, p2 = _f(xyz2)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    d1 = np.zeros((b, n), np.float32)
    d2 = np.zeros((b, m), np.float32)
    i1 = np.zeros((b, n), np.int32)
    i2 = np.zeros((b, m), np.int32)
    lib().oracle_chamfer_forward(b, n, m, p1, p2, d1.ctypes.data_as(_fp), d2.ctypes.data_as(_fp),
                                 i1.ctypes.data_as(_ip), i2.ctypes.data_as(_ip))
    return d1, d2, i1, i2


def chamfer_backward(xyz1, xyz2, idx1, idx2, g1, g2):
    xyz1, p1 = _f(xyz1)
    xyz2, p2 = _f(xyz2)
    idx1, pi1 = _i(idx1)
    idx2, pi2 = _i(idx2)
    g1, pg1 = _f(g1)
    g2, pg2 = _f(g2)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    o1 = np.zeros_like(xyz1)
    o2 = np.zeros_like(xyz2)
    lib().oracle_chamfer_backward(b, n, m, p1, p2, pi1, pi2, pg1, pg2, o1.ctypes.data_as(_fp), o2.ctypes.data_as(_fp))
    return o1, o2


def knn_generic(ref, query, k):
    """ref (dim,nr), query (dim,nq) -> dist (k,nq) fp32 (sqrt applied), ind (k,nq) int64 1-based."""
    ref, pr = _f(ref)
    query, pq = _f(query)
    dim, nr = ref.shape
    nq = query.shape[1]
    dist = np.zeros((k, nq), np.float32)
    ind = np.zeros((k, nq), np.int64)
    lib().oracle_knn_generic(pr, nr, pq, nq, dim, k, dist.ctypes.data_as(_fp), ind.ctypes.data_as(_lp))
    return dist, ind


def emd_forward(xyz1, xyz2, eps, iters, full_state=False):
    """-> (status, dist (b,n), assignment (b,n)); state initialised like emd_module.py:42-53."""
    xyz1, p1 = _f(xyz1)
    xyz2, p2 = _f(xyz2)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    dist = np.zeros((b, n), np.float32)
    assignment = np.full((b, n), -1, np.int32)
    assignment_inv = np.full((b, m), -1, np.int32)
    price = np.zeros((b, m), np.float32)
    bid = np.zeros((b, n), np.int32)
    bid_inc = np.zeros((b, n), np.float32)
    max_inc = np.zeros((b, m), np.float32)
    max_idx = np.zeros((b, m), np.int32)
    st = lib().oracle_emd_forward(b, n, m, p1, p2, dist.ctypes.data_as(_fp), assignment.ctypes.data_as(_ip),
                                  price.ctypes.data_as(_fp), assignment_inv.ctypes.data_as(_ip), bid.ctypes.data_as(_ip),
                                  bid_inc.ctypes.data_as(_fp), max_inc.ctypes.data_as(_fp), max_idx.ctypes.data_as(_ip),
                                  c_float(eps), int(iters))
    if full_state:
        return int(st), dist, assignment, {"price": price, "assignment_inv": assignment_inv, "bid": bid, "bid_increments": bid_inc,
                                           "max_increments": max_inc, "max_idx": max_idx}
    return int(st), dist, assignment


def emd_backward(xyz1, xyz2, grad_dist, assignment):
    xyz1, p1 = _f(xyz1)
    xyz2, p2 = _f(xyz2)
    grad_dist, pg = _f(grad_dist)
    assignment, pa = _i(assignment)
    b, n, _ = xyz1.shape
    out = np.zeros_like(xyz1)
    lib().oracle_emd_backward(b, n, p1, p2, pg, pa, out.ctypes.data_as(_fp))
    return out
