"""csrc/train_glue.hip -- the aggregation heads' small non-GEMM steps under autograd (NetVLAD soft-max / residual / intra-normalisation,
F.normalize, BatchNorm1d over a few rows, APFA attention) -- against the torch statements of the same steps evaluated in float64 on the CPU
(patch_aug_net/models/loupe.py:8-66, :196-222).  Values and every gradient; tolerance 2e-5 of the tensor's scale (fp32 sums, different order)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def close(got, ref, rtol=2e-5):
    got, ref = got.detach().double().cpu(), ref.detach().double().cpu()
    assert got.shape == ref.shape, (got.shape, ref.shape)
    scale = ref.abs().max().item() + 1e-12
    err = (got - ref).abs().max().item()
    assert err <= rtol * scale, f"max err {err:.3e} vs scale {scale:.3e}"


def run_both(fn_hip, fn_ref, inputs, seed=0):
    """inputs: list of CPU fp32 tensors; returns after comparing the value and the gradient of every input under a random cotangent."""
    hip_in = [t.detach().clone().cuda().requires_grad_(True) for t in inputs]
    ref_in = [t.detach().clone().double().requires_grad_(True) for t in inputs]
    out_h, out_r = fn_hip(*hip_in), fn_ref(*ref_in)
    close(out_h, out_r)
    w = torch.randn(out_r.shape, generator=torch.Generator().manual_seed(seed), dtype=torch.float64)
    (out_h * w.float().cuda()).sum().backward()
    (out_r * w).sum().backward()
    for i, (a, b) in enumerate(zip(hip_in, ref_in)):
        assert a.grad is not None, f"input {i} received no gradient"
        close(a.grad, b.grad)


@pytest.mark.parametrize("B,C,K,N", [(3, 32, 4, 128), (2, 256, 16, 1000), (2, 256, 64, 4096), (1, 8, 300, 70), (2, 16, 1, 64), (1, 6, 1100, 33), (2, 40, 7, 300)])
def test_netvlad_tail(B, C, K, N):
    from patchaugnet_amd import train_ops
    g = torch.Generator().manual_seed(B * 1000 + K)
    pre, x, cw2 = torch.randn(B, K, N, generator=g) * 2, torch.randn(B, C, N, generator=g), torch.randn(1, C, K, generator=g) / C ** 0.5

    def ref(pre, x, cw2):
        act = torch.softmax(pre, dim=1)
        a = act.sum(-1).unsqueeze(1) * cw2
        return F.normalize(torch.matmul(x, act.transpose(1, 2)) - a, dim=1, p=2)
    run_both(train_ops.netvlad_tail, ref, [pre, x, cw2])


@pytest.mark.parametrize("shape", [(5, 256), (3, 64, 1000), (2, 256, 1024), (1, 7, 3)])
def test_l2_normalize(shape):
    from patchaugnet_amd import train_ops
    x = torch.randn(*shape, generator=torch.Generator().manual_seed(len(shape)))
    run_both(train_ops.l2_normalize, lambda t: F.normalize(t, dim=1), [x])


def test_l2_normalize_of_a_zero_vector_is_zero_with_the_clamped_gradient():
    from patchaugnet_amd import train_ops
    x = torch.randn(4, 16)
    x[2] = 0
    run_both(train_ops.l2_normalize, lambda t: F.normalize(t, dim=1), [x])


@pytest.mark.parametrize("B,C,K", [(4, 32, 84), (18, 256, 84), (2, 16, 300), (1, 5, 1), (1, 3, 1100)])
def test_afa_attention(B, C, K):
    from patchaugnet_amd import train_ops
    g = torch.Generator().manual_seed(C + K)
    x, r = torch.randn(B, C, K, generator=g), torch.randn(B, C, K, generator=g)

    def ref(x, r):
        w = torch.softmax(r.max(dim=1)[0], dim=-1).unsqueeze(1)
        return F.relu(x + x * w)
    run_both(train_ops.afa_attention, ref, [x, r])


@pytest.mark.parametrize("R,Fdim", [(18, 256), (5, 40), (36, 1000)])
def test_bn_rows_train_matches_batchnorm1d(R, Fdim):
    from patchaugnet_amd import train_ops
    g = torch.Generator().manual_seed(R)
    bn_ref = torch.nn.BatchNorm1d(Fdim).double()
    with torch.no_grad():
        bn_ref.weight.copy_(torch.rand(Fdim, generator=g) + 0.5)
        bn_ref.bias.copy_(torch.randn(Fdim, generator=g) * 0.1)
        bn_ref.running_mean.copy_(torch.randn(Fdim, generator=g))
        bn_ref.running_var.copy_(torch.rand(Fdim, generator=g) + 0.5)
    bn_hip = torch.nn.BatchNorm1d(Fdim)
    bn_hip.load_state_dict({k: (v.float() if v.is_floating_point() else v) for k, v in bn_ref.state_dict().items()})
    bn_hip = bn_hip.cuda()
    x = torch.randn(R, Fdim, generator=g) * 3 + 1
    xh = x.clone().cuda().requires_grad_(True)
    xr = x.clone().double().requires_grad_(True)
    for _ in range(2):                                   # two steps: the running statistics and the counter move twice
        oh = train_ops.bn_rows(bn_hip, xh, True)
        orf = bn_ref(xr)
    close(oh, orf)
    w = torch.randn(R, Fdim, generator=g, dtype=torch.float64)
    xh.grad = None
    xr.grad = None
    bn_hip.zero_grad()
    bn_ref.zero_grad()
    (oh * w.float().cuda()).sum().backward()
    (orf * w).sum().backward()
    close(xh.grad, xr.grad)
    close(bn_hip.weight.grad, bn_ref.weight.grad)
    close(bn_hip.bias.grad, bn_ref.bias.grad)
    close(bn_hip.running_mean, bn_ref.running_mean)
    close(bn_hip.running_var, bn_ref.running_var)
    assert int(bn_hip.num_batches_tracked) == int(bn_ref.num_batches_tracked) == 2


def test_chain_train_counts_batches_inside_the_finalize_kernel():
    """num_batches_tracked of the chain's BatchNorm layers: += 1 per forward (+= B with per-cloud statistics groups), no library launch."""
    from patchaugnet_amd import train_ops
    bn = torch.nn.BatchNorm1d(16).cuda()
    W = torch.randn(16, 8, device="cuda", requires_grad=True)
    x = torch.randn(3, 8, 50, device="cuda")
    train_ops.chain_train(x, [train_ops.BNLayer(W, bn)], training=True)
    assert int(bn.num_batches_tracked) == 1
    train_ops.chain_train(x, [train_ops.BNLayer(W, bn)], groups=True, training=True)
    assert int(bn.num_batches_tracked) == 4
    bn.eval()          # every BatchNorm follows ITS OWN mode flag, like the torch module: a frozen layer inside a train() container keeps its statistics
    rm = bn.running_mean.clone()
    train_ops.chain_train(x, [train_ops.BNLayer(W, bn)], training=True)
    assert int(bn.num_batches_tracked) == 4 and torch.equal(bn.running_mean, rm)


def test_zero_arena_hands_out_zero_filled_disjoint_buffers_and_learns_its_size():
    """train_ops.zero_arena: the first step measures, later steps slice ONE filled buffer; escaped tensors stay valid after the step."""
    from patchaugnet_amd import _arena, train_ops
    dev = torch.device("cuda", torch.cuda.current_device())
    _arena._arena.demand.pop(dev, None)
    kept = []
    for step in range(3):
        with train_ops.zero_arena("cuda"):
            a = train_ops.zeros((5, 7), torch.float32, dev)
            b = train_ops.zeros((33,), torch.float64, dev)
            c = train_ops.zeros((2, 3, 4), torch.float32, dev)
            assert a.dtype == torch.float32 and b.dtype == torch.float64 and a.shape == (5, 7) and c.shape == (2, 3, 4)
            assert not a.any() and not b.any() and not c.any()
            shared = a.untyped_storage().data_ptr() == b.untyped_storage().data_ptr()
            assert shared == (step > 0), "the first step measures, the next ones share one buffer"
            a += 1
            b += 2
            c += 3
            assert (a == 1).all() and (b == 2).all() and (c == 3).all()          # disjoint
            big = train_ops.zeros((1 << 16,), torch.float32, dev)                 # beyond what was measured only in step 0 .. then learnt
            assert not big.any()
            kept.append(a)
    assert all((t == 1).all() for t in kept)                                      # an escaped tensor keeps its step's buffer alive
    for step in range(2):                                                         # keep=True (parameter gradients): a buffer of their own
        with train_ops.zero_arena("cuda"):
            big = train_ops.zeros((1 << 18,), torch.float32, dev)
            g = train_ops.zeros((16,), torch.float32, dev, keep=True)
            assert not g.any() and not big.any()
            if step:
                assert g.untyped_storage().data_ptr() != big.untyped_storage().data_ptr()
                assert g.untyped_storage().nbytes() < 4096                        # a kept gradient does not pin the step's large buffer
    assert train_ops.zeros((4,), torch.float32, dev).untyped_storage().nbytes() == 16   # outside a step: plain torch.zeros


def test_zero_arena_serves_backward_nodes_on_the_steps_stream_and_nobody_on_another_stream():
    """Autograd runs a device's backward nodes on its own worker thread, on the forward's stream: they are ordered after the step's fill launch
    and get slices of the step's buffer (the thread-bound rule of an earlier revision sent every backward accumulator to its own fill launch:
    +28 launches per training step).  A different STREAM is not ordered after the fill and gets a buffer of its own."""
    import threading
    from patchaugnet_amd import _arena, train_ops
    dev = torch.device("cuda", torch.cuda.current_device())
    _arena._arena.demand.pop(dev, None)
    seen = []

    class Twice(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            return x * 2

        @staticmethod
        def backward(ctx, g):
            z = train_ops.zeros((64,), torch.float32, g.device)
            seen.append((threading.get_ident(), z.untyped_storage().data_ptr(), bool(z.any())))
            return g * 2 + z[0]

    side = torch.cuda.Stream()
    for step in range(2):
        with train_ops.zero_arena("cuda"):
            base = train_ops.zeros((8,), torch.float32, dev)
            x = torch.ones(4, device=dev, requires_grad=True)
            Twice.apply(x).sum().backward()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                other = train_ops.zeros((8,), torch.float32, dev)
            torch.cuda.current_stream().wait_stream(side)
        assert not seen[-1][2] and not other.any() and (x.grad == 2).all()
        if step:
            assert seen[-1][1] == base.untyped_storage().data_ptr(), "a backward node on the step's stream shares the step's buffer"
            assert other.untyped_storage().data_ptr() != base.untyped_storage().data_ptr(), "another stream gets its own"


def test_interpolation_backward_with_prebuilt_lists_is_identical():
    """The inverted (point, neighbour) lists of interpolation's backward built ahead (backbone geometry(), prefetched with the neighbour
    searches) give the gradient the inline build gives (up to the order within a point's list), and the fp64 scatter-add's."""
    from patchaugnet_amd import pointops
    g = torch.Generator().manual_seed(4)
    b, c, n, m = 3, 32, 2048, 512
    unknown, known = torch.rand(b, n, 3, generator=g).cuda(), torch.rand(b, m, 3, generator=g).cuda()
    dist, idx = pointops.nearestneighbor(unknown, known)
    w = 1.0 / (dist + 1e-8)
    w = w / w.sum(2, keepdim=True)
    assert pointops._gather_form(b, c, n, m)
    lists = pointops.interpolation_backward_lists(idx, w, m)
    cot = torch.randn(b, c, n, generator=g).cuda()
    grads = []
    for l in (None, lists, lists):
        f = torch.randn(b, c, m, generator=torch.Generator().manual_seed(9)).cuda().requires_grad_(True)
        out = pointops.interpolation(f, idx, w, l)
        (out * cot).sum().backward()
        grads.append(f.grad)
    assert torch.equal(grads[1], grads[2])                   # the same lists: the same sums in the same order
    close(grads[0], grads[1].double(), rtol=1e-6)            # another build orders a point's list differently (atomic slot counters): rounding only
    ref = torch.zeros(b, c, m, dtype=torch.float64)
    idc, wc, cc = idx.cpu().long(), w.cpu().double(), cot.cpu().double()
    for j in range(3):
        ref.scatter_add_(2, idc[:, :, j].unsqueeze(1).expand(-1, c, -1), cc * wc[:, :, j].unsqueeze(1))
    close(grads[0], ref)
    # the interpolated features concatenated with skip features (backbone.py FPModule.forward): autograd hands the backward a channel SLICE of
    # the wider gradient; it is read in place through the batch stride (no contiguous copy) and gives the same sums
    f = torch.randn(b, c, m, generator=torch.Generator().manual_seed(9)).cuda().requires_grad_(True)
    skip = torch.randn(b, 5, n, generator=g).cuda()
    wide = torch.cat([pointops.interpolation(f, idx, w, lists), skip], dim=1)
    cot_wide = torch.cat([cot, torch.randn(b, 5, n, generator=g).cuda()], dim=1)
    (wide * cot_wide).sum().backward()
    assert torch.equal(f.grad, grads[1])


@pytest.mark.parametrize("wd", [0.0, 1e-2])
def test_hip_adam_matches_torch_adam(wd):
    """patchaugnet_amd.optim.Adam (csrc/adam.hip: the tensor list in the kernel arguments, a device step counter) against torch.optim.Adam on the
    same parameters and gradient sequence: ragged sizes (1 .. 1 000 003 elements, more than one launch's 84 tensors), five steps, and the
    state_dict moving both ways."""
    from patchaugnet_amd.optim import Adam
    g = torch.Generator().manual_seed(3)
    sizes = [1, 3, 5, 17, 255, 256, 257, 4095, 4096, 4097, 60 * 1024, 1000003] + [37 + 11 * i for i in range(90)]
    base = [torch.randn(n, generator=g) for n in sizes]
    pa = [torch.nn.Parameter(t.clone().cuda()) for t in base]
    pb = [torch.nn.Parameter(t.clone().cuda()) for t in base]
    oa = Adam(pa, lr=1e-2, betas=(0.9, 0.99), eps=1e-8, weight_decay=wd)
    ob = torch.optim.Adam(pb, lr=1e-2, betas=(0.9, 0.99), eps=1e-8, weight_decay=wd)
    for step in range(5):
        for x, y in zip(pa, pb):
            gr = torch.randn(x.shape, generator=g).cuda() * (0.1 + step)
            x.grad, y.grad = gr.clone(), gr.clone()
        oa.step(); ob.step()
    torch.cuda.synchronize()
    for x, y in zip(pa, pb):
        assert torch.allclose(x, y, rtol=2e-5, atol=2e-6), (x.numel(), (x - y).abs().max().item())
    sa = oa.state_dict()
    assert float(sa["state"][0]["step"]) == 5.0 and set(sa["state"][0]) == {"step", "exp_avg", "exp_avg_sq"}
    assert torch.allclose(sa["state"][5]["exp_avg"], ob.state_dict()["state"][5]["exp_avg"], rtol=2e-5, atol=1e-7)
    # a torch checkpoint continues on the HIP optimizer (and the counter is adopted)
    oc = Adam([torch.nn.Parameter(t.detach().clone()) for t in pb], lr=1e-2, betas=(0.9, 0.99), eps=1e-8, weight_decay=wd)
    import copy
    oc.load_state_dict(copy.deepcopy(ob.state_dict()))      # (load_state_dict keeps same-device tensors by reference: without the copy both optimizers would update ONE exp_avg)
    pc = oc.param_groups[0]["params"]
    for x, y, z in zip(pa, pb, pc):
        gr = torch.randn(x.shape, generator=g).cuda()
        y.grad, z.grad = gr.clone(), gr.clone()
    ob.step(); oc.step()
    for y, z in zip(pb, pc):
        assert torch.allclose(y, z, rtol=2e-5, atol=2e-6)
    with pytest.raises(RuntimeError):
        bad = torch.nn.Parameter(torch.zeros(4))
        bad.grad = torch.zeros(4)
        Adam([bad]).step()
    # ... and the other way: a HIP checkpoint through torch.save / torch.load into torch.optim.Adam.  The live state shares one
    # device counter per group; the checkpoint must not (torch's _foreach_add_ would advance a shared tensor once per parameter per step)
    import io
    sa = oa.state_dict()
    steps = [st["step"] for st in sa["state"].values()]
    assert len({t.data_ptr() for t in steps}) == len(steps), "state_dict() step tensors alias one storage"
    buf = io.BytesIO()
    torch.save(sa, buf)
    buf.seek(0)
    od = torch.optim.Adam([torch.nn.Parameter(t.detach().clone()) for t in pa], lr=1e-2, betas=(0.9, 0.99), eps=1e-8, weight_decay=wd)
    od.load_state_dict(torch.load(buf, weights_only=False))
    pd_ = od.param_groups[0]["params"]
    for x, w in zip(pa, pd_):
        gr = torch.randn(x.shape, generator=g).cuda()
        x.grad, w.grad = gr.clone(), gr.clone()
    oa.step(); od.step()
    torch.cuda.synchronize()
    assert all(float(st["step"]) == 6.0 for st in od.state_dict()["state"].values())
    assert float(oa.state_dict()["state"][0]["step"]) == 6.0
    for x, w in zip(pa, pd_):
        assert torch.allclose(x, w, rtol=2e-5, atol=2e-6), (x.numel(), (x - w).abs().max().item())
    # load_state_dict on an optimizer that has ALREADY stepped adopts the checkpoint's counter and learning rate (not its own stale ones)
    ck = copy.deepcopy(ob.state_dict())                      # torch optimizer at step 6, lr 1e-2
    ck["param_groups"][0]["lr"] = 5e-3
    oe = Adam([torch.nn.Parameter(t.detach().clone()) for t in pb], lr=1e-1, betas=(0.9, 0.99), eps=1e-8, weight_decay=wd)
    for z in oe.param_groups[0]["params"]:
        z.grad = torch.ones_like(z)
    oe.step(); oe.step()                                     # own counter at 2, own lr scalar at 1e-1
    pe = oe.param_groups[0]["params"]
    with torch.no_grad():
        for y, z in zip(pb, pe):
            z.copy_(y)
    oe.load_state_dict(ck)
    for gq in ob.param_groups:
        gq["lr"] = 5e-3
    for y, z in zip(pb, pe):
        gr = torch.randn(y.shape, generator=g).cuda()
        y.grad, z.grad = gr.clone(), gr.clone()
    ob.step(); oe.step()
    torch.cuda.synchronize()
    assert float(oe.state_dict()["state"][0]["step"]) == 7.0
    for y, z in zip(pb, pe):
        assert torch.allclose(y, z, rtol=2e-5, atol=2e-6), (y.numel(), (y - z).abs().max().item())


def test_hip_adam_follows_a_learning_rate_schedule_eagerly_and_under_graph_replay():
    """A learning-rate scheduler (train_place_recognition.py:531-568: StepLR / CosineAnnealingLR stepping per epoch) on patchaugnet_amd.optim.Adam:
    the learning rate is a device scalar, so (a) eager steps match torch.optim.Adam under the same StepLR, (b) a hipGraph captured around
    optimizer.step() applies a LATER learning rate once sync_hyperparameters() has run (what train.GraphedTrainer.step does before a replay) --
    with a launch-constant learning rate the replay would keep the captured one."""
    from patchaugnet_amd.optim import Adam
    g = torch.Generator().manual_seed(9)
    base = [torch.randn(n, generator=g) for n in (1000, 33, 4097)]
    pa = [torch.nn.Parameter(t.clone().cuda()) for t in base]
    pb = [torch.nn.Parameter(t.clone().cuda()) for t in base]
    oa, ob = Adam(pa, lr=1e-2), torch.optim.Adam(pb, lr=1e-2)
    sa, sb = torch.optim.lr_scheduler.StepLR(oa, step_size=2, gamma=0.1), torch.optim.lr_scheduler.StepLR(ob, step_size=2, gamma=0.1)
    grads = [[torch.randn(t.shape, generator=g).cuda() for t in base] for _ in range(6)]
    for step in range(6):
        for x, y, gr in zip(pa, pb, grads[step]):
            x.grad, y.grad = gr.clone(), gr.clone()
        oa.step(); ob.step(); sa.step(); sb.step()
    for x, y in zip(pa, pb):
        assert torch.allclose(x, y, rtol=2e-5, atol=2e-6)
    assert abs(oa.param_groups[0]["lr"] - 1e-5) < 1e-12
    # (b) capture one optimizer step on static gradients, then change lr and replay
    pc = [torch.nn.Parameter(t.clone().cuda()) for t in base]
    pd = [torch.nn.Parameter(t.clone().cuda()) for t in base]
    oc, od = Adam(pc, lr=1e-2), torch.optim.Adam(pd, lr=1e-2)
    for x, gr in zip(pc, grads[0]):
        x.grad = gr.clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        oc.step()                                       # state and device scalars exist before the capture
    torch.cuda.current_stream().wait_stream(side)
    gph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gph):
        oc.step()
    for x, gr in zip(pd, grads[0]):
        x.grad = gr.clone()
    od.step(); od.step()                                # the eager step + the captured one (capture does not execute; replay below does)
    gph.replay()
    oc.param_groups[0]["lr"] = od.param_groups[0]["lr"] = 1e-3
    oc.sync_hyperparameters()
    gph.replay()
    od.step()
    torch.cuda.synchronize()
    for x, y in zip(pc, pd):
        assert torch.allclose(x, y, rtol=2e-5, atol=2e-6), (x - y).abs().max().item()
