"""Generate tests/golden/*.npz by running the REFERENCE's own Python model classes.

Runs ONLY in the build container (needs /root/reference, read-only).  The
reference has no CPU implementation of its CUDA ops, so -- as SURVEY.md section 8c
prescribes -- the oracle op layer (oracle/pointops_cpu.py) is registered under
the module name the reference imports (``libs.pointops.functions.pointops``)
before its model packages are imported.  Everything above the op layer
(grouping modules' call order, SharedMLP, FP modules, NetVLAD pyramid, APFA,
decoder, state-dict key names) is then the reference's own code, unmodified.

One model family per process: the reference's two model packages both do
``import loupe`` as a top-level name.

Usage:  python -m oracle.gen_golden            (all families)
        python -m oracle.gen_golden patch_aug_net|pptnet|pointnet_vlad|keys
Nothing from the reference (source or bytecode) is written into this repo: the
outputs are input/output vectors and key/shape tables only.
"""
import json
import os
import subprocess
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
SEED_FWD = 4321          # torch.manual_seed before every forward (randperm in pointops.py:555)


def _inject():
    sys.dont_write_bytecode = True
    sys.path.insert(0, ROOT)
    sys.path.insert(1, REF)
    from oracle import pointops_cpu
    pkg = types.ModuleType("libs.pointops.functions")
    pkg.__path__ = []
    pkg.pointops = pointops_cpu
    sys.modules["libs.pointops.functions"] = pkg
    sys.modules["libs.pointops.functions.pointops"] = pointops_cpu
    return pointops_cpu


def _inputs(num_points=4096):
    from patchaugnet_amd.weights import synthetic_submaps
    return torch.cat([synthetic_submaps(1, num_points, 1234, "uniform"),
                      synthetic_submaps(1, num_points, 99, "street")], dim=0)


def _summ(t):
    """Small, order-sensitive summary of a large feature map: moments + strided samples."""
    t = t.detach().double().flatten()
    w = torch.linspace(0.5, 1.5, t.numel(), dtype=torch.float64)
    return np.array([t.mean(), t.abs().mean(), (t * w).sum() / t.numel(), t.min(), t.max()], dtype=np.float64)


def _samples(t, n=4096):
    f = t.detach().flatten()
    step = max(f.numel() // n, 1)
    return f[::step][:n].numpy().copy()


def gen_patch_aug_net():
    _inject()
    from patchaugnet_amd import configs
    from patchaugnet_amd.weights import seeded_state_dict
    from place_recognition.patch_aug_net.models import patch_aug_net as ref
    from oracle import models_cpu
    out = {}
    for tag, cfg, npts in (("full", configs.patch_aug_net_config(), 4096),
                           ("small", configs.scaled_config(configs.patch_aug_net_config(), 512), 512)):
        model = ref.Network(param=cfg, use_a2a_recon=True, use_l2_norm=True)
        sd = seeded_state_dict(model.state_dict())
        model.load_state_dict(sd, strict=True)
        model.eval()
        x = _inputs(npts)
        with torch.no_grad():
            torch.manual_seed(SEED_FWD)
            desc, fp, cidx = model(x)
            torch.manual_seed(SEED_FWD)
            res = model.backbone(x.squeeze(1))
            # training-path call: origin patches + decoder on cloud 0 / 1 (patch_aug_net.py:68-104)
            torch.manual_seed(SEED_FWD)
            (desc2, recon), _, _ = model(x, nn_dict={(0, 1): None})
            # the restatement must reproduce the reference classes on the same ops
            torch.manual_seed(SEED_FWD)
            d_or, fp_or, c_or = models_cpu.patch_aug_net_forward(sd, cfg, x)
        assert torch.equal(desc, desc2)
        err = (d_or - desc).abs().max().item()
        print(f"[patch_aug_net/{tag}] oracle-vs-reference max|d desc| = {err:.3e}")
        assert err <= 1e-6, err
        for a, b in zip(c_or, cidx):
            assert torch.equal(a, b)
        out.update({
            f"{tag}_x": x.numpy(),
            f"{tag}_desc": desc.numpy(),
            **{f"{tag}_center_idx{i}": c.numpy() for i, c in enumerate(cidx)},
            **{f"{tag}_sample_idx{i}": s.numpy() for i, s in enumerate(res["sample_idx_origin"])},
            **{f"{tag}_fp{i}_summary": _summ(f) for i, f in enumerate(fp)},
            **{f"{tag}_fp{i}_samples": _samples(f) for i, f in enumerate(fp)},
            # NB res["sa_features"] aliases the list the FP loop overwrites (patch_aug_net.py:167,
            # :183-187), so only its last entry is a true SA output; keep that one.
            f"{tag}_sa2_summary": _summ(res["sa_features"][2]),
            f"{tag}_sa2_samples": _samples(res["sa_features"][2]),
            f"{tag}_recon0_samples": _samples(recon["reconstructed_patches"][0]),
            f"{tag}_recon0_summary": _summ(recon["reconstructed_patches"][0]),
            f"{tag}_origin_patches0_samples": _samples(recon["origin_patches"][0]),
        })
        if tag == "full":
            keys = {k: [list(v.shape), str(v.dtype)] for k, v in model.state_dict().items()}
            with open(os.path.join(GOLD, "patch_aug_net_state_dict_keys.json"), "w") as f:
                json.dump(keys, f, indent=0)
    out["seed_fwd"] = np.array(SEED_FWD)
    np.savez_compressed(os.path.join(GOLD, "patch_aug_net.npz"), **out)


def gen_pptnet():
    _inject()
    from patchaugnet_amd import configs
    from patchaugnet_amd.weights import seeded_state_dict
    from place_recognition.pptnet_origin.models import pptnet as ref
    from oracle import models_cpu
    out = {}
    for tag, cfg, npts in (("full", configs.pptnet_config(), 4096),
                           ("small", configs.scaled_config(configs.pptnet_config(), 1024), 1024)):
        for norm in (False, True):
            model = ref.Network(param=cfg, use_normalize=norm)
            sd = seeded_state_dict(model.state_dict())
            model.load_state_dict(sd, strict=True)
            model.eval()
            x = _inputs(npts)
            with torch.no_grad():
                desc, fp, cidx = model(x)
                d_or, fp_or, c_or = models_cpu.pptnet_forward(sd, cfg, x, use_normalize=norm)
            err = ((d_or - desc).abs().max() / desc.abs().max()).item()
            print(f"[pptnet/{tag}/norm={norm}] oracle-vs-reference rel max|d desc| = {err:.3e}")
            assert err <= 1e-6, err
            sfx = "_l2" if norm else ""
            out[f"{tag}_desc{sfx}"] = desc.numpy()
        out.update({
            f"{tag}_x": x.numpy(),
            **{f"{tag}_center_idx{i}": c.numpy() for i, c in enumerate(cidx)},
            **{f"{tag}_fp{i}_summary": _summ(f) for i, f in enumerate(fp)},
            **{f"{tag}_fp{i}_samples": _samples(f) for i, f in enumerate(fp)},
        })
        if tag == "full":
            keys = {k: [list(v.shape), str(v.dtype)] for k, v in model.state_dict().items()}
            with open(os.path.join(GOLD, "pptnet_state_dict_keys.json"), "w") as f:
                json.dump(keys, f, indent=0)
    np.savez_compressed(os.path.join(GOLD, "pptnet.npz"), **out)


def gen_pointnet_vlad():
    """BASELINE.json configs[0]: PointNetVlad as evaluate.py:88-90 builds it; the reference class runs on CPU unmodified."""
    sys.dont_write_bytecode = True
    sys.path.insert(0, ROOT)
    sys.path.insert(1, REF)
    from patchaugnet_amd.weights import seeded_state_dict
    from place_recognition.pointnet_vlad import PointNetVlad as ref
    out = {}
    for tag, npts in (("full", 4096), ("small", 512)):
        model = ref.PointNetVlad(global_feat=True, feature_transform=True, max_pool=False, output_dim=256, num_points=npts)
        sd = seeded_state_dict(model.state_dict())
        model.load_state_dict(sd, strict=True)
        model.eval()
        x = _inputs(npts)
        with torch.no_grad():
            desc = model(x)
            d1 = model(x[:1])                                   # batch = 1, the configuration BASELINE.json names
        assert torch.allclose(d1, desc[:1], atol=2e-5), (d1 - desc[:1]).abs().max()
        out[f"{tag}_x"] = x.numpy()
        out[f"{tag}_desc"] = desc.numpy()
        if tag == "full":
            keys = {k: [list(v.shape), str(v.dtype)] for k, v in model.state_dict().items()}
            with open(os.path.join(GOLD, "pointnet_vlad_state_dict_keys.json"), "w") as f:
                json.dump(keys, f, indent=0)
        print(f"[pointnet_vlad/{tag}] desc {tuple(desc.shape)} |desc|max {desc.abs().max():.3f}")
    np.savez_compressed(os.path.join(GOLD, "pointnet_vlad.npz"), **out)


def main():
    os.makedirs(GOLD, exist_ok=True)
    which = sys.argv[1:] or ["patch_aug_net", "pptnet", "pointnet_vlad"]
    if len(which) > 1:
        for w in which:
            subprocess.check_call([sys.executable, "-m", "oracle.gen_golden", w], cwd=ROOT,
                                  env=dict(os.environ, PYTHONDONTWRITEBYTECODE="1"))
        return
    {"patch_aug_net": gen_patch_aug_net, "pptnet": gen_pptnet, "pointnet_vlad": gen_pointnet_vlad}[which[0]]()


if __name__ == "__main__":
    main()
