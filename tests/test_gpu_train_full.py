"""BASELINE.json configs[3] at FULL size on the MI355X, against numbers produced by the reference's own code.

tests/golden/train_step.npz (oracle/gen_train_golden.py, build container): the reference's patch_aug_net.Network in train() mode on one
18-cloud x 4096-point tuple with nn_dict (train_place_recognition.py:142-164), its quadruplet_loss + patch_chamfer_loss
(losses/pointnetvlad_loss.py:53-105, :242-247) and loss.backward() -- descriptors, both loss values, per-parameter gradient norms and
samples, BatchNorm running statistics after the step.  The product runs the same step through patchaugnet_amd.train on the GPU.

Also here: EMD at the reference's call shape (-1, 4096, 3), eps 0.02, 1024 iterations (pointnetvlad_loss.py:205-221).
"""
import numpy as np
import pytest
import torch

from tests._util import golden

pytestmark = pytest.mark.gpu


def _model():
    from patchaugnet_amd import configs, patch_aug_net
    from patchaugnet_amd.weights import seeded_state_dict
    m = patch_aug_net.Network(param=configs.patch_aug_net_config(), use_a2a_recon=True, use_l2_norm=True)
    m.load_state_dict(seeded_state_dict(m.state_dict()), strict=True)
    return m.cuda()


def _inputs(z):
    from oracle.gen_train_golden import tuple_inputs
    x = tuple_inputs()
    assert np.array_equal(x[0, 0, :8].numpy(), z["x_head"])
    assert np.allclose([x.double().sum().item(), x.double().abs().sum().item()], z["x_checksum"], rtol=0, atol=1e-6)
    return x


def test_full_size_training_step_matches_reference_run():
    from oracle.gen_train_golden import ARGS, NN_DICT
    from patchaugnet_amd import losses
    z = golden("train_step")
    m = _model().train()
    x = _inputs(z).cuda().requires_grad_(True)
    torch.manual_seed(int(z["seed_fwd"]))
    (desc, recon), _, _ = m(x, NN_DICT)
    assert list(recon["cloud_indices"]) == z["cloud_indices"].tolist()
    err = (desc.detach().cpu() - torch.from_numpy(z["desc"])).abs().max().item()
    assert err <= 2e-4, f"train-mode descriptors differ from the reference run by {err:.3e}"
    d = desc.view(1, -1, 256)
    oq, op, on, oo = torch.split(d, [1, 2, 14, 1], dim=1)
    place = losses.quadruplet_loss(oq, op, on, oo, ARGS["m1"], ARGS["m2"], use_min=ARGS["use_min"], lazy=ARGS["lazy"], ignore_zero_loss=ARGS["ignore_zero_loss"])
    rec = losses.patch_chamfer_loss(recon["origin_patches"], recon["reconstructed_patches"])
    assert abs(place.item() - float(z["loss_place"])) <= 2e-4 * max(1.0, abs(float(z["loss_place"]))), (place.item(), float(z["loss_place"]))
    assert abs(rec.item() - float(z["loss_recon"])) <= 2e-4 * max(1.0, abs(float(z["loss_recon"]))), (rec.item(), float(z["loss_recon"]))
    (place + rec).backward()
    got = dict(m.named_parameters())
    bad = []
    for k in z["grad_names"].tolist():
        g = got[k].grad
        assert g is not None, k
        ref_n = float(z["gnorm/" + k])
        n = g.double().norm().item()
        f = g.detach().flatten().cpu()
        step = max(f.numel() // 8, 1)
        samp = f[::step][:8].numpy()
        tol_s = 1e-2 * max(ref_n / np.sqrt(f.numel()), 1e-7) + 1e-2 * np.abs(z["gsamp/" + k])
        # biases in front of a BatchNorm have a mathematically zero gradient: both sides hold rounding noise there (norms ~1e-6)
        if ref_n < 2e-5:
            assert n < 1e-4, (k, n, ref_n)
            continue
        if abs(n - ref_n) > 5e-3 * ref_n + 2e-5 or (np.abs(samp - z["gsamp/" + k]) > tol_s + 2e-5 / np.sqrt(f.numel())).any():
            bad.append((k, n, ref_n, float(np.abs(samp - z["gsamp/" + k]).max())))
    assert not bad, bad[:6]
    sd = m.state_dict()
    for k in [k for k in z.files if k.startswith("bnstat/")]:
        a = sd[k[len("bnstat/"):]].cpu().numpy()
        assert np.allclose(a, z[k], rtol=2e-4, atol=2e-5), k


def test_emd_at_the_reference_call_shape():
    """emd_loss (pointnetvlad_loss.py:205-221): (16, 4096, 3), eps 0.02, 1024 iterations; every cloud's assignment and squared
    distances bit-exact vs the oracle auction, and the loss value mean(mean(sqrt(dist))) equal to 1e-6."""
    from oracle import oracle_ops as o
    from patchaugnet_amd import emd_module
    rng = np.random.default_rng(5)
    a = rng.uniform(-1, 1, (16, 4096, 3)).astype(np.float32)
    c = (a[:, rng.permutation(4096)] + rng.normal(scale=0.05, size=a.shape)).astype(np.float32)
    x1, x2 = torch.from_numpy(a).cuda(), torch.from_numpy(c).cuda()
    dist, ass = emd_module.emdModule()(x1, x2, 0.02, 1024)
    torch.cuda.synchronize()
    st, rd, ra = o.emd_forward(a, c, 0.02, 1024)
    assert st == 1
    assert np.array_equal(ass.cpu().numpy(), ra)
    assert np.array_equal(dist.cpu().numpy(), rd)
    assert (ra >= 0).all() and min(len(np.unique(r)) for r in ra) >= 4090          # the auction has (all but) converged at this eps
    loss_hip = torch.sqrt(dist).mean(1).mean().item()
    loss_ref = float(np.sqrt(rd.astype(np.float64)).mean(1).mean())
    assert abs(loss_hip - loss_ref) <= 1e-6 * max(1.0, loss_ref)
