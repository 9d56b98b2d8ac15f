"""The grouping MODULES of the op layer on the HIP path against the CPU oracle's restatement of the same classes
(libs/pointops/functions/pointops.py:476-516 QueryAndGroup ball + kNN, :519-582 QueryAndGroup_Edge, :584-635 QueryAndGroup_Edge_Split,
:637-661 GroupAll).  Neighbour lists are index outputs -> bit-exact; the grouped tensors are pure gathers and one fp32 subtraction in
the reference's order -> bit-exact as well; the feature gradients are sums of atomics -> 1e-6 relative."""
import numpy as np
import pytest
import torch

from oracle import pointops_cpu as C

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def P():
    from patchaugnet_amd import pointops
    return pointops


def _inputs(b, n, m, c, seed, lattice=False):
    g = torch.Generator().manual_seed(seed)
    xyz = torch.rand(b, n, 3, generator=g) * 2 - 1
    if lattice:                       # exact distance ties: the (d2, index) order decides
        xyz = torch.round(xyz * 4) / 4
    sel = torch.stack([torch.randperm(n, generator=g)[:m] for _ in range(b)])
    new_xyz = torch.gather(xyz, 1, sel.unsqueeze(-1).expand(-1, -1, 3)).contiguous()
    feats = torch.randn(b, c, n, generator=g)
    cfeat = torch.gather(feats, 2, sel.unsqueeze(1).expand(-1, c, -1)).contiguous()
    return xyz, new_xyz, feats, cfeat


def _same(a, b):
    assert a.shape == b.shape and a.dtype == b.dtype, (a.shape, b.shape, a.dtype, b.dtype)
    assert torch.equal(a, b), f"max|diff| {(a.double() - b.double()).abs().max().item():.3e}"


CASES = [(2, 512, 64, 16, 20, False), (1, 1000, 37, 5, 8, True), (3, 128, 128, 32, 32, False), (2, 64, 16, 3, 1, False)]


@pytest.mark.parametrize("b,n,m,c,ns,lattice", CASES)
@pytest.mark.parametrize("radius", [None, 0.35])
@pytest.mark.parametrize("use_xyz", [True, False])
def test_query_and_group_ball_and_knn(P, b, n, m, c, ns, lattice, radius, use_xyz):
    xyz, new_xyz, feats, _ = _inputs(b, n, m, c, seed=100 + n + ns, lattice=lattice)
    ref = C.QueryAndGroup(radius, ns, use_xyz)(xyz, new_xyz, feats)
    got = P.QueryAndGroup(radius, ns, use_xyz)(xyz.cuda(), new_xyz.cuda(), feats.cuda())
    _same(got.cpu(), ref)
    # features=None -> centred coordinates only; new_xyz=None -> the cloud queries itself (pointops.py:492-493)
    ref = C.QueryAndGroup(radius, ns, True)(xyz, None, None)
    got = P.QueryAndGroup(radius, ns, True)(xyz.cuda(), None, None)
    _same(got.cpu(), ref)
    # a caller-supplied neighbour list bypasses the search
    idx = (C.ballquery(radius, ns, xyz, new_xyz) if radius is not None else C.knnquery(ns, xyz, new_xyz))
    got = P.QueryAndGroup(radius, ns, use_xyz)(xyz.cuda(), new_xyz.cuda(), feats.cuda(), idx.cuda())
    _same(got.cpu(), C.QueryAndGroup(radius, ns, use_xyz)(xyz, new_xyz, feats, idx))


def test_query_and_group_needs_xyz_or_features(P):
    xyz, new_xyz, _, _ = _inputs(1, 64, 8, 4, seed=3)
    with pytest.raises(AssertionError):
        P.QueryAndGroup(None, 4, use_xyz=False)(xyz.cuda(), new_xyz.cuda(), None)


@pytest.mark.parametrize("b,n,m,c,ns,lattice", CASES)
@pytest.mark.parametrize("radius", [None, 0.35])
@pytest.mark.parametrize("ret_gxyz", [False, True])
def test_query_and_group_edge_split(P, b, n, m, c, ns, lattice, radius, ret_gxyz):
    xyz, new_xyz, feats, cfeat = _inputs(b, n, m, c, seed=200 + n + ns, lattice=lattice)
    r_feat, r_xyz = C.QueryAndGroup_Edge_Split(radius, ns, True, ret_gxyz)(xyz, new_xyz, feats, cfeat)
    g_feat, g_xyz = P.QueryAndGroup_Edge_Split(radius, ns, True, ret_gxyz)(xyz.cuda(), new_xyz.cuda(), feats.cuda(), cfeat.cuda())
    _same(g_feat.cpu(), r_feat)
    _same(g_xyz.cpu(), r_xyz)
    r_feat, r_xyz = C.QueryAndGroup_Edge_Split(radius, ns, False, ret_gxyz)(xyz, new_xyz, feats, cfeat)
    g_feat, g_xyz = P.QueryAndGroup_Edge_Split(radius, ns, False, ret_gxyz)(xyz.cuda(), new_xyz.cuda(), feats.cuda(), cfeat.cuda())
    _same(g_feat.cpu(), r_feat)
    _same(g_xyz.cpu(), r_xyz)
    r_feat, r_xyz = C.QueryAndGroup_Edge_Split(radius, ns, True, ret_gxyz)(xyz, new_xyz, None, None)
    g_feat, g_xyz = P.QueryAndGroup_Edge_Split(radius, ns, True, ret_gxyz)(xyz.cuda(), new_xyz.cuda(), None, None)
    _same(g_feat.cpu(), r_feat)
    _same(g_xyz.cpu(), r_xyz)


@pytest.mark.parametrize("radius", [None, 0.4])
def test_query_and_group_edge_ball_and_returns(P, radius):
    """QueryAndGroup_Edge outside the configuration the models use: ball neighbourhoods, no dilation, every return combination."""
    xyz, new_xyz, feats, cfeat = _inputs(2, 300, 40, 12, seed=77)
    for ret_gxyz in (False, True):
        for ret_idx in (False, True):
            ref = C.QueryAndGroup_Edge(radius, 16, 1, True, ret_gxyz, ret_idx)(xyz, new_xyz, feats, cfeat)
            got = P.QueryAndGroup_Edge(radius, 16, 1, True, ret_gxyz, ret_idx)(xyz.cuda(), new_xyz.cuda(), feats.cuda(), cfeat.cuda())

            def flat(r):
                out = []
                while isinstance(r, tuple):
                    out.append(r[1])
                    r = r[0]
                return [r] + out[::-1]
            for a, e in zip(flat(got), flat(ref)):
                _same(a.cpu(), e)


def test_query_and_group_edge_dilated_matches_reference_permutation(P):
    """knn_dilation > 1: the reference searches dilation x nsample and keeps columns randperm(nsample) (pointops.py:553-555); the same
    CPU-generator draw selects the same columns here."""
    xyz, new_xyz, feats, cfeat = _inputs(2, 512, 64, 8, seed=5)
    torch.manual_seed(123)
    r_feat, r_idx = C.QueryAndGroup_Edge(None, 20, 2, True, False, True)(xyz, new_xyz, feats, cfeat)
    torch.manual_seed(123)
    g_feat, g_idx = P.QueryAndGroup_Edge(None, 20, 2, True, False, True)(xyz.cuda(), new_xyz.cuda(), feats.cuda(), cfeat.cuda())
    _same(g_idx.cpu(), r_idx)
    _same(g_feat.cpu(), r_feat)


@pytest.mark.parametrize("use_xyz", [True, False])
def test_group_all(P, use_xyz):
    xyz, new_xyz, feats, _ = _inputs(3, 200, 1, 24, seed=9)
    _same(P.GroupAll(use_xyz)(xyz.cuda(), new_xyz.cuda(), feats.cuda()).cpu(), C.GroupAll(use_xyz)(xyz, new_xyz, feats))
    _same(P.GroupAll(use_xyz)(xyz.cuda(), new_xyz.cuda(), None).cpu(), C.GroupAll(use_xyz)(xyz, new_xyz, None))


@pytest.mark.parametrize("mod", ["QueryAndGroup", "QueryAndGroup_Edge_Split", "GroupAll"])
def test_group_module_feature_gradients(P, mod):
    """Backward through the modules: the feature gradient is the grouping backward (grouping_cuda_kernel.cu:33-52), the centre-feature
    gradient the plain sum over the neighbourhood; coordinates carry no gradient into the neighbour search."""
    xyz, new_xyz, feats, cfeat = _inputs(2, 256, 32, 10, seed=31)

    def run(M, dev):
        f = feats.detach().clone().to(dev).requires_grad_(True)          # fresh leaves per run (.to("cpu") alone would alias the input)
        cf = cfeat.detach().clone().to(dev).requires_grad_(True)
        if mod == "QueryAndGroup":
            out = M.QueryAndGroup(None, 12, True)(xyz.to(dev), new_xyz.to(dev), f)
        elif mod == "QueryAndGroup_Edge_Split":
            out = M.QueryAndGroup_Edge_Split(0.5, 12, True)(xyz.to(dev), new_xyz.to(dev), f, cf)[0]
        else:
            out = M.GroupAll(True)(xyz.to(dev), new_xyz.to(dev), f)
        w = torch.randn(out.shape, generator=torch.Generator().manual_seed(8)).to(dev)
        (out * w).sum().backward()
        return f.grad.cpu(), (cf.grad.cpu() if cf.grad is not None else None)

    rf, rc = run(C, "cpu")
    gf, gc = run(P, "cuda")
    assert torch.allclose(gf, rf, rtol=1e-6, atol=1e-6), (gf - rf).abs().max()
    assert (gc is None) == (rc is None)
    if rc is not None:
        assert torch.allclose(gc, rc, rtol=1e-5, atol=1e-5), (gc - rc).abs().max()


@pytest.mark.parametrize("attention", [False, True])
def test_multi_scale_grouping_level_against_the_oracle(attention):
    """backbone.SAModuleMSG (PointNet2SAModuleMSG, patch_aug_net.py:246-287 + the base forward :203-243): one sampling, two scales with their own
    neighbour counts and shared MLPs, features concatenated along the channels and neighbour lists along the last axis -- on the HIP op layer and
    MFMA dense kernels against the same statement assembled from the CPU oracle's ops and a float64 copy of the same layers."""
    from patchaugnet_amd.backbone import SAModule, SAModuleMSG
    from patchaugnet_amd.weights import seeded_state_dict
    b, n, m, c = 2, 256, 32, 16
    xyz, _, feats, _ = _inputs(b, n, m, c, seed=77)
    msg = SAModuleMSG(npoint=m, radii=[None, None], nsamples=[8, 16], knn_dilation=1, mlps=[[c, 16, 32], [c, 32, 64]], gp=4, attention=attention)
    msg.load_state_dict(seeded_state_dict(msg.state_dict(), seed=5))
    assert [k for k in msg.state_dict() if k.startswith("mlps.1.")] and (not attention or [k for k in msg.state_dict() if k.startswith("sas.1.")])
    msg = msg.eval()
    # oracle composition on the CPU: FPS -> gather -> per scale (EdgeConv group, float64 MLP + max [, attention])
    cidx = C.furthestsampling(xyz, m)
    new_xyz = torch.gather(xyz, 1, cidx.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
    cfeat = torch.gather(feats, 2, cidx.long().unsqueeze(1).expand(-1, c, -1)).contiguous()
    ref_f, ref_i = [], []
    cpu = SAModuleMSG(npoint=m, radii=[None, None], nsamples=[8, 16], knn_dilation=1, mlps=[[c, 16, 32], [c, 32, 64]], gp=4, attention=attention)
    cpu.load_state_dict(msg.state_dict())
    cpu = cpu.double().eval()
    with torch.no_grad():
        for i, ns in enumerate([8, 16]):
            grouped, sidx = C.QueryAndGroup_Edge(None, ns, knn_dilation=1, use_xyz=True, ret_sample_idx=True)(xyz, new_xyz, feats, cfeat)
            y = cpu.mlps[i](grouped.double()).max(dim=3)[0]
            if attention:
                y = cpu.sas[i](y)
            ref_f.append(y)
            ref_i.append(sidx)
        got = msg.cuda()(xyz.cuda(), feats.cuda())
    nx, ci, si, f = got
    assert torch.equal(ci.cpu(), cidx) and torch.equal(nx.cpu(), new_xyz)
    assert torch.equal(si.cpu(), torch.cat(ref_i, dim=-1))
    ref = torch.cat(ref_f, dim=1)
    assert f.shape == (b, 32 + 64, m)
    assert (f.cpu().double() - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item())
    # one scale through the MSG class == the single-scale level
    one = SAModuleMSG(npoint=m, radii=[None], nsamples=[8], knn_dilation=1, mlps=[[c, 16, 32]])
    single = SAModule(mlp=[c, 16, 32], npoint=m, nsample=8)
    one.load_state_dict(seeded_state_dict(one.state_dict(), seed=9))
    single.load_state_dict(one.state_dict())
    with torch.no_grad():
        a, bb_ = one.eval().cuda()(xyz.cuda(), feats.cuda()), single.eval().cuda()(xyz.cuda(), feats.cuda())
    assert all(torch.equal(u, v) for u, v in zip(a, bb_))


@pytest.mark.parametrize("b,c,n,m,k", [(3, 64, 1024, 128, 20), (2, 256, 128, 16, 20), (1, 5, 37, 7, 3), (2, 16, 5000, 33, 16)])
def test_edge_group_fused_op_equals_the_unfused_statement(b, c, n, m, k):
    """pointops.EdgeGroup (csrc/group_edge.hip: gathering + grouping + centre subtraction + cat with the grouped coordinates as one launch each
    way) against the statement it replaces (pointops.py:559-570 spelled with pa_gathering / pa_grouping and torch subtract / cat under autograd):
    the forward bit for bit (a gather and one subtraction), the feature gradient to fp32 summation order."""
    from patchaugnet_amd import pointops as P
    g = torch.Generator().manual_seed(b * 1000 + n)
    feats = torch.randn(b, c, n, generator=g).cuda()
    cidx = torch.stack([torch.randperm(n, generator=g)[:m] for _ in range(b)]).int().cuda()
    idx = torch.randint(0, n, (b, m, k), generator=g).int().cuda()
    idx[:, :, 0] = cidx                                         # a centre is its own first neighbour, as in the model
    gxyz = torch.randn(b, 3, m, k, generator=g).cuda()
    go = torch.randn(b, 3 + c, m, k, generator=g).cuda()
    f1 = feats.clone().requires_grad_(True)
    out1 = P.edge_group(f1, cidx, idx, gxyz)
    out1.backward(go)
    f0 = feats.clone().requires_grad_(True)
    out0 = torch.cat([gxyz, P.grouping(f0, idx) - P.gathering(f0, cidx).unsqueeze(-1)], dim=1)
    out0.backward(go)
    assert torch.equal(out1, out0)
    scale = f0.grad.abs().max().item()
    assert (f1.grad - f0.grad).abs().max().item() <= 1e-5 * max(scale, 1.0)


@pytest.mark.parametrize("b,n,m,k", [(3, 4096, 1024, 20), (2, 128, 16, 20), (1, 37, 7, 3)])
def test_coordinate_only_geometry_ops_equal_the_unfused_statements(b, n, m, k):
    """The one-launch forms the training loop's geometry prefetch uses, against the statements they replace: furthestsampling + gathering
    (patch_aug_net.py:222-225), grouping of the coordinates minus the centre (pointops.py:559-562; twice along the channel axis = the first level's
    cat with its own features), the inverse-distance weights (patch_aug_net.py:350-353) and torch.gather on the index lists (:169-177)."""
    from patchaugnet_amd import pointops as P
    g = torch.Generator().manual_seed(n)
    xyz = (torch.rand(b, n, 3, generator=g) * 2 - 1).cuda()
    ci, nx = P.furthestsampling_gather(xyz, m)
    ci0 = P.furthestsampling(xyz, m)
    nx0 = P.gathering(xyz.transpose(1, 2).contiguous(), ci0).transpose(1, 2).contiguous()
    assert torch.equal(ci, ci0) and torch.equal(nx, nx0)
    idx = P.knnquery(k, xyz, nx)
    og0 = P.grouping(xyz.transpose(1, 2).contiguous(), idx)
    cen0 = og0 - nx.transpose(1, 2).unsqueeze(-1)
    og, cen = P.grouped_coordinates(xyz, nx, idx)
    assert torch.equal(og, og0) and torch.equal(cen, cen0)
    og2, cen2 = P.grouped_coordinates_fused(xyz, nx, idx, 2)
    feats = xyz.transpose(1, 2).contiguous()
    first = torch.cat([cen0, P.grouping(feats, idx) - P.gathering(feats, ci).unsqueeze(-1)], dim=1)
    assert torch.equal(og2, og0) and torch.equal(cen2, first)
    i3, w3 = P.three_nn_weights(xyz, nx)
    d, i0 = P.nearestneighbor(xyz, nx)
    r = 1.0 / (d + 1e-8)
    w0 = r / torch.sum(r, dim=2, keepdim=True)
    assert torch.equal(i3, i0) and (w3 - w0).abs().max().item() <= 2e-7
    table = torch.randint(0, 10 ** 6, (b, n), generator=g).int().cuda()
    for ix in (ci, idx):
        ref = torch.gather(table.unsqueeze(1).expand(-1, ix.shape[1], -1), -1, ix.long()) if ix.dim() == 3 else torch.gather(table, -1, ix.long())
        got = P.compose_indices(table, ix)
        assert got.dtype == torch.int32 and torch.equal(got, ref)


def test_sampled_centres_keep_their_gradient_path_when_the_coordinates_need_one():
    """furthestsampling_gather is one no-gradient launch for plain inputs, but with coordinates that require a gradient (train.run_model(...,
    input_grad=True): the reference's feed.requires_grad_, train_place_recognition.py:155) the sampled centres must stay differentiable: they
    enter the centred neighbour coordinates.  d sum(new_xyz) / d xyz = 1 on the sampled rows, 0 elsewhere; same indices either way."""
    from patchaugnet_amd import pointops as P
    g = torch.Generator().manual_seed(2)
    xyz = (torch.rand(3, 500, 3, generator=g) * 2 - 1).cuda()
    ci0, nx0 = P.furthestsampling_gather(xyz, 64)
    assert nx0.grad_fn is None and not nx0.requires_grad
    xg = xyz.clone().requires_grad_(True)
    ci, nx = P.furthestsampling_gather(xg, 64)
    assert torch.equal(ci, ci0) and torch.equal(nx.detach(), nx0) and nx.requires_grad
    nx.sum().backward()
    want = torch.zeros_like(xyz)
    want.scatter_(1, ci.long().unsqueeze(-1).expand(-1, -1, 3), 1.0)
    assert torch.equal(xg.grad, want)
    with torch.no_grad():
        _, nx2 = P.furthestsampling_gather(xg, 64)
    assert not nx2.requires_grad
