"""The drop-in boundary EXECUTED on the MI355X.

  * group 2 of include/patchaugnet_hip.h: every launcher symbol the reference's torch bindings link against
    (libs/pointops/src/*/..._cuda_kernel.h), called through ctypes with raw device pointers exactly as the reference's *_cuda.cpp
    files call them (sizes first, null stream where the original has none), each result compared with the CPU oracle;
  * patchaugnet_amd.pointops_cuda: every function of the reference's native module `pointops_cuda` (pointops_api.cpp:15-40) with
    the reference's calling convention -- the CALLER allocates the outputs (pointops.py:20-22, :43-44, :132-133, :424-426, ...).
Index outputs and gathers bit-exact; the backward scatters to 1e-6 (float atomics, like the reference's atomicAdd).
"""
import ctypes

import numpy as np
import pytest
import torch

from oracle import oracle_ops as o

pytestmark = pytest.mark.gpu
RNG = np.random.default_rng(21)
I, F, P = ctypes.c_int, ctypes.c_float, ctypes.c_void_p


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def p(t):
    return P(t.data_ptr())


_ALIVE = []


def _keep(t):
    """Hold a device tensor until the test ends: raw pointers handed to the C ABI do not keep their tensor alive."""
    _ALIVE.append(t)
    return t


def zeros(*s, dt=torch.float32):
    return torch.zeros(*s, dtype=dt, device="cuda")


@pytest.fixture(scope="module")
def lib():
    from patchaugnet_amd import _lib
    l = ctypes.CDLL(_lib.LIB_PATH)
    return l


def _sig(fn, spec):
    fn.argtypes = [{"i": I, "f": F, "p": P}[c] for c in spec]
    fn.restype = None
    return fn


def _data():
    b, n, m, k, c = 2, 700, 90, 12, 10
    x = (RNG.random((b, n, 3), dtype=np.float32) * 2 - 1)
    q = (RNG.random((b, m, 3), dtype=np.float32) * 2 - 1)
    f = RNG.standard_normal((b, c, n)).astype(np.float32)
    return b, n, m, k, c, x, q, f


def test_group2_launchers_execute_and_match_the_oracle(lib):
    b, n, m, k, c, x, q, f = _data()
    X, Q, Fd = dev(x), dev(q), dev(f)
    S = torch.cuda.current_stream().cuda_stream
    torch.cuda.synchronize()        # group-2 launchers without a stream argument use the null stream, like the originals

    # K1 furthestsampling_cuda_launcher(b, n, m, dataset, temp, idxs)                       sampling_cuda_kernel.h:17
    temp, idx = torch.full((b, n), 1e10, device="cuda"), zeros(b, m, dt=torch.int32)
    _sig(lib.furthestsampling_cuda_launcher, "iiippp")(b, n, m, p(X), p(temp), p(idx))
    torch.cuda.synchronize()
    cidx = o.furthestsampling(x, m)
    assert np.array_equal(idx.cpu().numpy(), cidx)

    # K2/K3 gathering                                                                        sampling_cuda_kernel.h:15-16
    out = zeros(b, c, m)
    _sig(lib.gathering_forward_cuda_launcher, "iiiippp")(b, c, n, m, p(Fd), p(idx), p(out))
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), o.gathering_forward(f, cidx))
    g = RNG.standard_normal((b, c, m)).astype(np.float32)
    gp = zeros(b, c, n)
    _sig(lib.gathering_backward_cuda_launcher, "iiiippp")(b, c, n, m, p(_keep(dev(g))), p(idx), p(gp))
    torch.cuda.synchronize()
    assert np.allclose(gp.cpu().numpy(), o.gathering_backward(g, cidx, n), atol=1e-6)

    # K4 knnquery_cuda_launcher(b, n, m, nsample, xyz, new_xyz, idx, dist2, stream)           knnquery_cuda_kernel.h:14
    kidx, kd2 = zeros(b, m, k, dt=torch.int32), zeros(b, m, k)
    _sig(lib.knnquery_cuda_launcher, "iiiippppp")(b, n, m, k, p(X), p(Q), p(kidx), p(kd2), P(S))
    torch.cuda.synchronize()
    ri, rd = o.knnquery(k, x, q)
    assert np.array_equal(kidx.cpu().numpy(), ri) and np.array_equal(kd2.cpu().numpy(), rd)

    # K5/K6/K8 grouping                                                                        grouping_cuda_kernel.h:16-19
    for name in ("grouping_forward_cuda_launcher", "grouping_forward_cuda_launcher_fast"):
        go = zeros(b, c, m, k)
        _sig(getattr(lib, name), "iiiiippp")(b, c, n, m, k, p(Fd), p(kidx), p(go))
        torch.cuda.synchronize()
        assert np.array_equal(go.cpu().numpy(), o.grouping_forward(f, ri)), name
    gg = RNG.standard_normal((b, c, m, k)).astype(np.float32)
    gp = zeros(b, c, n)
    _sig(lib.grouping_backward_cuda_launcher, "iiiiippp")(b, c, n, m, k, p(_keep(dev(gg))), p(kidx), p(gp))
    torch.cuda.synchronize()
    assert np.allclose(gp.cpu().numpy(), o.grouping_backward(gg, ri, n), atol=1e-5)
    fi = RNG.integers(-2**40, 2**40, (b, 3, n), dtype=np.int64)
    for name in ("grouping_int_forward_cuda_launcher", "grouping_int_forward_cuda_launcher_fast"):
        gi = zeros(b, 3, m, k, dt=torch.int64)
        _sig(getattr(lib, name), "iiiiippp")(b, 3, n, m, k, p(_keep(dev(fi))), p(kidx), p(gi))
        torch.cuda.synchronize()
        assert np.array_equal(gi.cpu().numpy(), o.grouping_int_forward(fi, ri)), name

    # K9/K10/K11 three-NN + interpolation (unknown = the n points, known = the m queries)      interpolation_cuda_kernel.h:18-23
    rd2, ri3 = o.nearestneighbor(x, q)
    for name in ("nearestneighbor_cuda_launcher", "nearestneighbor_cuda_launcher_fast"):
        d2, i3 = zeros(b, n, 3), zeros(b, n, 3, dt=torch.int32)
        _sig(getattr(lib, name), "iiipppp")(b, n, m, p(X), p(Q), p(d2), p(i3))
        torch.cuda.synchronize()
        assert np.array_equal(i3.cpu().numpy(), ri3) and np.array_equal(d2.cpu().numpy(), rd2), name
    fk = RNG.standard_normal((b, c, m)).astype(np.float32)
    w = RNG.random((b, n, 3), dtype=np.float32)
    w /= w.sum(-1, keepdims=True)
    FK, W = dev(fk), dev(w)                      # named: a temporary would be freed (and its block re-used) before the launch reads it
    for name in ("interpolation_forward_cuda_launcher", "interpolation_forward_cuda_launcher_fast"):
        io = zeros(b, c, n)
        _sig(getattr(lib, name), "iiiipppp")(b, c, m, n, p(FK), p(i3), p(W), p(io))
        torch.cuda.synchronize()
        assert np.array_equal(io.cpu().numpy(), o.interpolation_forward(fk, ri3, w)), name
    gi_ = RNG.standard_normal((b, c, n)).astype(np.float32)
    gk, GI = zeros(b, c, m), dev(gi_)
    _sig(lib.interpolation_backward_cuda_launcher, "iiiipppp")(b, c, n, m, p(GI), p(i3), p(W), p(gk))     # positions (b, c, n, m)
    torch.cuda.synchronize()
    assert np.allclose(gk.cpu().numpy(), o.interpolation_backward(gi_, ri3, w, m), atol=1e-4)

    # K13 ball query: (new_xyz, xyz) order                                                      ballquery_cuda_kernel.h:15-17
    rb = o.ballquery(0.4, k, x, q)
    bi = zeros(b, m, k, dt=torch.int32)
    _sig(lib.ballquery_cuda_launcher, "iiifippp")(b, n, m, 0.4, k, p(Q), p(X), p(bi))
    torch.cuda.synchronize()
    assert np.array_equal(bi.cpu().numpy(), rb)
    bi = zeros(b, m, k, dt=torch.int32)
    _sig(lib.ballquery_cuda_launcher_fast, "iiifipppp")(b, n, m, 0.4, k, p(Q), p(X), p(bi), P(S))
    torch.cuda.synchronize()
    assert np.array_equal(bi.cpu().numpy(), rb)

    # K14/K15 featuredistribute / featuregather                                                 featuredistribute_cuda_kernel.h:15-17
    di = zeros(b, m, dt=torch.int32)
    _sig(lib.featuredistribute_cuda_launcher, "iiipppp")(b, n, m, p(X), p(Q), p(di), P(S))
    torch.cuda.synchronize()
    rdi = o.featuredistribute(x, q)
    assert np.array_equal(di.cpu().numpy(), rdi)
    fo = zeros(b, c, m)
    _sig(lib.featuregather_forward_cuda_launcher, "iiiipppp")(b, n, m, c, p(Fd), p(di), p(fo), P(S))
    torch.cuda.synchronize()
    assert np.array_equal(fo.cpu().numpy(), o.gathering_forward(f, rdi))
    gf = RNG.standard_normal((b, c, m)).astype(np.float32)
    gm = zeros(b, c, n)
    _sig(lib.featuregather_backward_cuda_launcher, "iiiipppp")(b, n, m, c, p(_keep(dev(gf))), p(di), p(gm), P(S))
    torch.cuda.synchronize()
    assert np.allclose(gm.cpu().numpy(), o.gathering_backward(gf, rdi, n), atol=1e-6)

    # K16 label statistics                                                                       labelstat_cuda_kernel.h:20-27
    ncls = 6
    lab = RNG.integers(0, 5, (b, n, ncls), dtype=np.int32)
    L = dev(lab)
    ra, rbi = o.labelstat_and_ballquery(0.4, k, x, q, lab)
    li, ls = zeros(b, m, k, dt=torch.int32), zeros(b, m, ncls, dt=torch.int32)
    _sig(lib.labelstat_and_ballquery_cuda_launcher_fast, "iiifiippppp" + "p")(b, n, m, 0.4, k, ncls, p(Q), p(X), p(L), p(li), p(ls), P(S))
    torch.cuda.synchronize()
    assert np.array_equal(ls.cpu().numpy(), ra) and np.array_equal(li.cpu().numpy(), rbi)
    ls = zeros(b, m, ncls, dt=torch.int32)
    _sig(lib.labelstat_ballrange_cuda_launcher_fast, "iiifipppp" + "p")(b, n, m, 0.4, ncls, p(Q), p(X), p(L), p(ls), P(S))
    torch.cuda.synchronize()
    assert np.array_equal(ls.cpu().numpy(), o.labelstat_ballrange(0.4, x, q, lab))
    ls = zeros(b, m, ncls, dt=torch.int32)
    _sig(lib.labelstat_idx_cuda_launcher_fast, "iiiiippp" + "p")(b, n, m, k, ncls, p(L), p(kidx), p(ls), P(S))
    torch.cuda.synchronize()
    assert np.array_equal(ls.cpu().numpy(), o.labelstat_idx(lab, ri))


def test_pointops_cuda_module_functions_execute_and_match_the_oracle():
    """The reference's `import pointops_cuda` surface with its pre-allocated-output convention (pointops_api.cpp:15-40): the product's
    mirror module on the C ABI."""
    from patchaugnet_amd import pointops_cuda as pc
    _exercise_pointops_cuda(pc)
    import inspect
    public = {n for n, fn in inspect.getmembers(pc, inspect.isfunction) if n.endswith("_cuda")}
    assert public == CALLED, public ^ CALLED


def test_reference_built_binding_layer_runs_on_this_library():
    """INTEGRATION.md route 2 on hardware: the reference's OWN pointops_api.cpp + */*_cuda.cpp, compiled in the build container against
    libpatchaugnet_hip.so (oracle/build_ref.py -> oracle/_ref/, test infrastructure), imported as `pointops_cuda` and driven exactly as
    the reference's pointops.py drives it.  Every launch therefore enters this repo's HIP kernels through the reference's launcher names."""
    import glob
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sos = glob.glob(os.path.join(root, "oracle", "_ref", "pointops_cuda*.so"))
    if not sos:
        pytest.skip("oracle/_ref/pointops_cuda*.so not built (needs the reference tree: python -m oracle.build_ref)")
    spec = importlib.util.spec_from_file_location("pointops_cuda", sos[0])
    pc = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(pc)
    assert CALLED <= set(dir(pc))
    _exercise_pointops_cuda(pc)


CALLED = {"furthestsampling_cuda", "gathering_forward_cuda", "gathering_backward_cuda", "knnquery_cuda", "grouping_forward_cuda",
          "grouping_backward_cuda", "grouping_int_forward_cuda", "nearestneighbor_cuda", "interpolation_forward_cuda",
          "interpolation_backward_cuda", "ballquery_cuda", "featuredistribute_cuda", "featuregather_forward_cuda",
          "featuregather_backward_cuda", "labelstat_and_ballquery_cuda", "labelstat_ballrange_cuda", "labelstat_idx_cuda"}


def _exercise_pointops_cuda(pc):
    b, n, m, k, c, x, q, f = _data()
    X, Q, Fd = dev(x), dev(q), dev(f)
    temp, idx = torch.full((b, n), 1e10, device="cuda"), zeros(b, m, dt=torch.int32)            # pointops.py:20-22
    pc.furthestsampling_cuda(b, n, m, X, temp, idx)
    cidx = o.furthestsampling(x, m)
    assert np.array_equal(idx.cpu().numpy(), cidx)
    out = zeros(b, c, m)                                                                         # :43-44
    pc.gathering_forward_cuda(b, c, n, m, Fd, idx, out)
    assert np.array_equal(out.cpu().numpy(), o.gathering_forward(f, cidx))
    g = RNG.standard_normal((b, c, m)).astype(np.float32)
    gp = zeros(b, c, n)                                                                          # :52-54
    pc.gathering_backward_cuda(b, c, n, m, dev(g), idx, gp)
    assert np.allclose(gp.cpu().numpy(), o.gathering_backward(g, cidx, n), atol=1e-6)
    kidx, kd2 = zeros(b, m, k, dt=torch.int32), zeros(b, m, k)                                   # :424-426
    pc.knnquery_cuda(b, n, m, k, X, Q, kidx, kd2)
    ri, rd = o.knnquery(k, x, q)
    assert np.array_equal(kidx.cpu().numpy(), ri) and np.array_equal(kd2.cpu().numpy(), rd)
    go = zeros(b, c, m, k)                                                                       # :132-133
    pc.grouping_forward_cuda(b, c, n, m, k, Fd, kidx, go)
    assert np.array_equal(go.cpu().numpy(), o.grouping_forward(f, ri))
    gg = RNG.standard_normal((b, c, m, k)).astype(np.float32)
    gp = zeros(b, c, n)                                                                          # :145-147
    pc.grouping_backward_cuda(b, c, n, m, k, dev(gg), kidx, gp)
    assert np.allclose(gp.cpu().numpy(), o.grouping_backward(gg, ri, n), atol=1e-5)
    fi = RNG.integers(-2**40, 2**40, (b, 3, n), dtype=np.int64)
    gi = zeros(b, 3, m, k, dt=torch.int64)                                                       # :163-165
    pc.grouping_int_forward_cuda(b, 3, n, m, k, dev(fi), kidx, gi)
    assert np.array_equal(gi.cpu().numpy(), o.grouping_int_forward(fi, ri))
    d2, i3 = zeros(b, n, 3), zeros(b, n, 3, dt=torch.int32)                                      # :73-75
    pc.nearestneighbor_cuda(b, n, m, X, Q, d2, i3)
    rd2, ri3 = o.nearestneighbor(x, q)
    assert np.array_equal(i3.cpu().numpy(), ri3) and np.array_equal(d2.cpu().numpy(), rd2)
    fk = RNG.standard_normal((b, c, m)).astype(np.float32)
    w = RNG.random((b, n, 3), dtype=np.float32)
    w /= w.sum(-1, keepdims=True)
    io = zeros(b, c, n)                                                                          # :100-102
    pc.interpolation_forward_cuda(b, c, m, n, dev(fk), i3, dev(w), io)
    assert np.array_equal(io.cpu().numpy(), o.interpolation_forward(fk, ri3, w))
    gi_ = RNG.standard_normal((b, c, n)).astype(np.float32)
    gk = zeros(b, c, m)                                                                          # :113-115
    pc.interpolation_backward_cuda(b, c, n, m, dev(gi_), i3, dev(w), gk)
    assert np.allclose(gk.cpu().numpy(), o.interpolation_backward(gi_, ri3, w, m), atol=1e-4)
    bi = zeros(b, m, k, dt=torch.int32)                                                          # :188-190 (new_xyz, xyz)
    pc.ballquery_cuda(b, n, m, 0.4, k, Q, X, bi)
    assert np.array_equal(bi.cpu().numpy(), o.ballquery(0.4, k, x, q))
    di = zeros(b, m, dt=torch.int32)                                                             # :212-214
    pc.featuredistribute_cuda(b, n, m, X, Q, di)
    rdi = o.featuredistribute(x, q)
    assert np.array_equal(di.cpu().numpy(), rdi)
    fo = zeros(b, c, m)                                                                          # :236-238
    pc.featuregather_forward_cuda(b, n, m, c, Fd, di, fo)
    assert np.array_equal(fo.cpu().numpy(), o.gathering_forward(f, rdi))
    gf = RNG.standard_normal((b, c, m)).astype(np.float32)
    gm = zeros(b, c, n)                                                                          # :251-253
    pc.featuregather_backward_cuda(b, n, m, c, dev(gf), di, gm)
    assert np.allclose(gm.cpu().numpy(), o.gathering_backward(gf, rdi, n), atol=1e-6)
    ncls = 6
    lab = RNG.integers(0, 5, (b, n, ncls), dtype=np.int32)
    L = dev(lab)
    ra, rbi = o.labelstat_and_ballquery(0.4, k, x, q, lab)
    li, ls = zeros(b, m, k, dt=torch.int32), zeros(b, m, ncls, dt=torch.int32)                   # :333-336
    pc.labelstat_and_ballquery_cuda(b, n, m, 0.4, k, ncls, Q, X, L, li, ls)
    assert np.array_equal(ls.cpu().numpy(), ra) and np.array_equal(li.cpu().numpy(), rbi)
    ls = zeros(b, m, ncls, dt=torch.int32)                                                       # :302-304
    pc.labelstat_ballrange_cuda(b, n, m, 0.4, ncls, Q, X, L, ls)
    assert np.array_equal(ls.cpu().numpy(), o.labelstat_ballrange(0.4, x, q, lab))
    ls = zeros(b, m, ncls, dt=torch.int32)                                                       # :275-277
    pc.labelstat_idx_cuda(b, n, m, k, ncls, L, kidx, ls)
    assert np.array_equal(ls.cpu().numpy(), o.labelstat_idx(lab, ri))
    torch.cuda.synchronize()
