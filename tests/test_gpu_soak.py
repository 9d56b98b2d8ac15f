"""Concurrency soaks: the two product faults of rounds 3-5 -- memset nodes of a captured training step that were not ordered in
front of the kernels accumulating into the buffer, and first-level sampling disturbed by fp16 MFMA kernels of OTHER streams -- were found by
multi-stream / many-replay runs that lived under tools/probes/.  These are their bounded forms (the whole file: under a minute on the MI355X):

  * four-stream replay soak: extract.GraphedExtractor on FOUR streams (the headline pipeline), both models x both MLP dtypes, a different batch per
    replay, eager launches / allocations / read-backs between replays in every second round; every replay must equal the serial forward of its
    batch BIT FOR BIT (the serial forward is what tests/test_gpu_models.py holds to the reference's vectors);
  * captured-training soak: train.GraphedTrainer(prefetch=True) + patchaugnet_amd.optim.Adam, >= 50 replays with a NEW tuple each and eager work in
    between; at four replays spread over the run the gradients the replay produced are compared with an EAGER training step on the same tuple,
    the same weights (snapshot taken before the replay) and the same kNN permutation; all gradients, weights and optimizer state stay finite.

Against the pre-fix library these fail (round 5: hipMemsetAsync nodes in the patch-Chamfer backward; operand-modifier forms of packed fp32 in the
sampling round): see DESIGN.md section 5.
"""
import copy

import pytest
import torch

from patchaugnet_amd.weights import seeded_state_dict, synthetic_submaps

pytestmark = pytest.mark.gpu


def _model(name):
    from patchaugnet_amd import configs, patch_aug_net, pptnet
    if name == "pptnet":
        m = pptnet.Network(param=configs.pptnet_config(), use_normalize=True)
    else:
        m = patch_aug_net.Network(param=configs.patch_aug_net_config(), use_a2a_recon=True, use_l2_norm=True)
    m.load_state_dict(seeded_state_dict(m.state_dict()))
    return m.cuda().eval()


@pytest.mark.parametrize("name,dtype", [("patch_aug_net", "f32"), ("patch_aug_net", "f16"), ("pptnet", "f32"), ("pptnet", "f16")])
def test_four_stream_replay_soak(name, dtype):
    from patchaugnet_amd.extract import GraphedExtractor
    m = _model(name)
    m.mlp_dtype = dtype
    nb, rounds = 12, 4
    xs = [synthetic_submaps(32, 4096, 170 + i, "street" if i % 3 == 0 else "uniform").cuda() for i in range(nb)]
    with torch.no_grad():
        ref = [m(x, return_feat=False).clone() for x in xs]
    gx = GraphedExtractor(m, (32, 1, 4096, 3), n_streams=4)
    out = torch.empty(nb, 32, 256, device="cuda")
    scratch = torch.empty(1 << 20, device="cuda")
    bad = []
    for r in range(rounds):
        out.fill_(float("nan"))
        gx.begin()
        for i in range(nb):
            j = (i * 5 + r) % nb                   # another order every round: a stream's slot sees another batch than last time
            gx.run(xs[j], out=out[j])
            if r % 2:                              # eager work between replays (what any real loop does): a launch, allocations, a read-back
                scratch.fill_(float(i))
                junk = torch.full((1 << 21,), float("nan"), device="cuda")
                float(scratch[0])
                del junk
        gx.end()
        torch.cuda.synchronize()
        bad += [(r, i, float((out[i] - ref[i]).abs().max())) for i in range(nb) if not torch.equal(out[i], ref[i])]
    assert not bad, f"{len(bad)} of {rounds * nb} replays on four streams differ from the serial forward: {bad[:6]}"


def test_captured_training_soak_gradients_against_eager_steps():
    from patchaugnet_amd import configs, patch_aug_net, pointops
    from patchaugnet_amd.optim import Adam
    from patchaugnet_amd.train import DEFAULTS, GraphedTrainer, training_step
    n, steps, checks = 1024, 52, (1, 17, 33, 51)             # odd replays: p.grad references the buffer set captured last (set 1)
    cfg = configs.scaled_config(configs.patch_aug_net_config(), n)
    m = patch_aug_net.Network(param=cfg, use_a2a_recon=True, use_l2_norm=True)
    m.load_state_dict(seeded_state_dict(m.state_dict()))
    m = m.cuda()
    twin = copy.deepcopy(m)
    g = torch.Generator().manual_seed(9)

    def new_tuple():
        base = torch.rand(1, 1, n, 3, generator=g) * 2 - 1
        q = base + 0.01 * torch.randn(1, 1, n, 3, generator=g)
        pos = base + 0.02 * torch.randn(1, 2, n, 3, generator=g)
        neg, oth = torch.rand(1, 4, n, 3, generator=g) * 2 - 1, torch.rand(1, 1, n, 3, generator=g) * 2 - 1
        return tuple(t.cuda() for t in (q, pos, neg, oth))
    nn_dict = {(0, 1): None, (0, 2): None}
    args = dict(DEFAULTS, TRAIN_NEGATIVES_PER_QUERY=4)
    opt = Adam(m.parameters(), lr=1e-4)
    cur = new_tuple()
    tr = GraphedTrainer(m, opt, *cur, nn_dict, num_points=n, args=args, warmup=2, prefetch=True)
    assert tr.prefetch
    twin_groupers = [x for x in twin.modules() if isinstance(x, pointops.QueryAndGroup_Edge) and x.radius is None and x.knn_dilation > 1]
    assert len(twin_groupers) == len(tr.groupers)
    sgd0 = torch.optim.SGD(twin.parameters(), lr=0.0)
    scratch = torch.empty(1 << 20, device="cuda")
    names = [k for k, _ in m.named_parameters()]
    recon = []
    for i in range(steps):
        nxt = new_tuple()                                                    # fresh device allocations every step
        scratch.fill_(float(i))
        junk = [torch.full((1 << 19,), float("nan"), device="cuda") for _ in range(4)]
        float(scratch[0])
        del junk
        if i in checks:
            torch.cuda.synchronize()
            snap = copy.deepcopy(m.state_dict())
        k = tr._k
        losses = tr.step(*cur, next_batch=nxt)
        if i in checks:
            assert k == 1
            torch.cuda.synchronize()
            got = {nm: p.grad.clone() for nm, p in m.named_parameters() if p.grad is not None}
            twin.load_state_dict(snap)
            for gr, hp in zip(twin_groupers, tr._perm_host[k]):
                gr.perm_buffer = hp.cuda()
            training_step(twin, sgd0, *cur, nn_dict=nn_dict, num_points=n, args=args)
            for gr in twin_groupers:
                gr.perm_buffer = None
            torch.cuda.synchronize()
            want = {nm: p.grad for nm, p in twin.named_parameters() if p.grad is not None}
            assert set(got) == set(want)
            for nm in names:
                if nm in want:
                    a, b = want[nm].double(), got[nm].double()
                    assert torch.isfinite(b).all(), (i, nm)
                    # same kernels, another order of the fp32 atomics (split-K weight gradients): relative L2 per tensor
                    assert (a - b).norm().item() <= 2e-2 * max(a.norm().item(), 1e-3), (i, nm, a.norm().item(), (a - b).norm().item())
        if i % 10 == 0 or i == steps - 1:
            recon.append(float(losses["patch_recon_a2a"]))
        cur = nxt
    torch.cuda.synchronize()
    assert all(bool(torch.isfinite(v).all()) for v in m.state_dict().values() if v.is_floating_point()), "weights went non-finite during the soak"
    assert all(bool(torch.isfinite(t).all()) for st in opt.state.values() for t in st.values() if torch.is_tensor(t)), "optimizer state went non-finite"
    assert steps <= float(opt.state_dict()["state"][0]["step"]) <= steps + 4      # the soak's replays + the trainer's warm-up steps: every replay ticked the device counter
    assert all(r == r and r < 10.0 for r in recon), recon
    tr.close()
