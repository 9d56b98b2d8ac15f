"""Full-size training-step fixture (BASELINE.json configs[3]) from the REFERENCE's own model and loss code (build container only).

What runs, unmodified, from /root/reference:
  * place_recognition/patch_aug_net/models/patch_aug_net.py  Network in train() mode (BatchNorm batch statistics), called like
    run_model does (train_place_recognition.py:142-164): one tuple of 18 clouds (1 query + 2 positives + 14 negatives + 1 other
    negative, configs/patch_aug_net.yaml:60-62) of 4096 points, nn_dict = the two (query, positive) pairs -> 3 related clouds;
  * losses/pointnetvlad_loss.py  quadruplet_loss (:53-105, the YAML's training arguments) and patch_chamfer_loss (:242-247) through
    libs/chamfer_dist/__init__.py (ChamferFunction / ChamferDistanceL1).
Native modules the reference imports by name are registered from the oracle first: the op layer
``libs.pointops.functions.pointops`` (oracle/pointops_cpu.py, autograd included) and ``chamfer`` (oracle_chamfer_* of
oracle/pointops_oracle.c); ``emd`` is an empty stub (not called).  loss.backward() then gives the reference's gradients.

Stored (tests/golden/train_step.npz): the two loss values, the 18 train-mode descriptors, the L2 norm and eight strided samples of
every parameter's gradient, and the BatchNorm running statistics after the step for two layers.  Inputs are re-made from seeds.

Usage: python -m oracle.gen_train_golden
"""
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
SEED_FWD = 4321
NUM_POINTS = 4096
NN_DICT = {(0, 1): None, (0, 2): None}
ARGS = dict(m1=0.5, m2=0.2, use_min=False, lazy=True, ignore_zero_loss=False)      # configs/patch_aug_net.yaml:55-75


def tuple_inputs(num_points=NUM_POINTS):
    """(18, 1, N, 3): query, 2 positives (the query's scene, re-sampled), 14 negatives + 1 other negative (other scenes)."""
    from oracle.gen_e2e_golden import submap
    pl = [0, 0, 0] + list(range(1, 16))
    return torch.from_numpy(np.stack([submap(9000 + p, 9100 + i, num_points) for i, p in enumerate(pl)])).unsqueeze(1)


def _chamfer_module():
    from oracle import oracle_ops as o
    m = types.ModuleType("chamfer")

    def forward(xyz1, xyz2):
        d1, d2, i1, i2 = o.chamfer_forward(xyz1.detach().numpy(), xyz2.detach().numpy())
        return torch.from_numpy(d1), torch.from_numpy(d2), torch.from_numpy(i1), torch.from_numpy(i2)

    def backward(xyz1, xyz2, idx1, idx2, g1, g2):
        a, b = o.chamfer_backward(xyz1.detach().numpy(), xyz2.detach().numpy(), idx1.numpy(), idx2.numpy(),
                                  g1.contiguous().numpy(), g2.contiguous().numpy())
        return torch.from_numpy(a), torch.from_numpy(b)
    m.forward, m.backward = forward, backward
    return m


def grad_samples(g, n=8):
    f = g.detach().flatten()
    step = max(f.numel() // n, 1)
    return f[::step][:n].numpy().copy()


def main():
    from oracle.gen_golden import _inject
    _inject()
    sys.modules["chamfer"] = _chamfer_module()
    sys.modules.setdefault("emd", types.ModuleType("emd"))
    from patchaugnet_amd import configs
    from patchaugnet_amd.weights import seeded_state_dict
    from place_recognition.patch_aug_net.models import patch_aug_net as ref
    from losses import pointnetvlad_loss as L
    cfg = configs.patch_aug_net_config()
    model = ref.Network(param=cfg, use_a2a_recon=True, use_l2_norm=True)
    model.load_state_dict(seeded_state_dict(model.state_dict()), strict=True)
    model.train()
    x = tuple_inputs().requires_grad_(True)
    torch.manual_seed(SEED_FWD)
    (desc, recon), _, _ = model(x, NN_DICT)
    d = desc.view(1, -1, 256)
    oq, op, on, oo = torch.split(d, [1, 2, 14, 1], dim=1)
    place = L.quadruplet_loss(oq, op, on, oo, ARGS["m1"], ARGS["m2"], use_min=ARGS["use_min"], lazy=ARGS["lazy"], ignore_zero_loss=ARGS["ignore_zero_loss"])
    rec = L.patch_chamfer_loss(recon["origin_patches"], recon["reconstructed_patches"])
    (place + rec).backward()
    out = {"seed_fwd": np.array(SEED_FWD), "x_head": x.detach()[0, 0, :8].numpy(),
           "x_checksum": np.array([x.detach().double().sum().item(), x.detach().double().abs().sum().item()]),
           "desc": desc.detach().numpy(), "loss_place": np.array(place.item()), "loss_recon": np.array(rec.item()),
           "cloud_indices": np.array(recon["cloud_indices"]), "recon0_samples": grad_samples(recon["reconstructed_patches"][0], 64)}
    names = []
    for k, p in model.named_parameters():
        if p.grad is None:
            continue
        names.append(k)
        out["gnorm/" + k] = np.array(p.grad.double().norm().item())
        out["gsamp/" + k] = grad_samples(p.grad)
    out["grad_names"] = np.array(names)
    sd = model.state_dict()
    for k in ("backbone.SA_modules.0.mlps.0.layer0.bn.bn.running_mean", "backbone.FP_modules.0.mlp.layer2.bn.bn.running_var",
              "aggregation.vlads.2.bn1.running_mean", "decoder.bn1.running_var"):
        out["bnstat/" + k] = sd[k].numpy().copy()
    print("loss place %.6f recon %.6f; %d parameters with gradients; |grad x| %.4e" % (place.item(), rec.item(), len(names), x.grad.norm().item()))
    np.savez_compressed(os.path.join(GOLD, "train_step.npz"), **out)
    print("wrote train_step.npz", os.path.getsize(os.path.join(GOLD, "train_step.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
