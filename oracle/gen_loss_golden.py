"""Generate tests/golden/losses.npz from the REFERENCE's own loss functions and loader (build container only).

``losses/pointnetvlad_loss.py`` imports the native ``chamfer`` / ``emd`` modules at import time; they are stubbed with empty
modules (the descriptor losses dumped here are pure torch and never touch them).  ``utils/loading_pointclouds.py`` imports
cleanly.  Inputs are regenerated from seeds by ``loss_inputs`` / ``cloud_inputs`` below; only results are stored.

Usage: python -m oracle.gen_loss_golden
"""
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")

LOSS_CASES = [  # name, kwargs
    ("triplet_loss", dict(use_min=False, lazy=False, ignore_zero_loss=False)),
    ("triplet_loss", dict(use_min=True, lazy=True, ignore_zero_loss=True)),
    ("quadruplet_loss", dict(use_min=False, lazy=True, ignore_zero_loss=False, soft_margin=False)),     # train defaults
    ("quadruplet_loss", dict(use_min=False, lazy=False, ignore_zero_loss=True, soft_margin=False)),
    ("quadruplet_loss", dict(use_min=True, lazy=False, ignore_zero_loss=False, soft_margin=True)),
    ("contrastive_quadruplet_loss", dict(use_min=False, lazy=True, ignore_zero_loss=False)),
    ("hphn_quadruplet_loss", dict()),
]


def loss_inputs(seed=7, b=4, p=2, nn=14, d=256, dtype=torch.float64):
    g = torch.Generator().manual_seed(seed)
    mk = lambda *s: torch.nn.functional.normalize(torch.randn(*s, generator=g, dtype=dtype), dim=-1)
    q = mk(b, 1, d)
    pos = torch.nn.functional.normalize(q + 0.6 * mk(b, p, d), dim=-1)
    spread = torch.tensor([0.45, 0.7, 1.0, 1.6], dtype=dtype)[:b].view(b, 1, 1)      # some rows have negatives closer than positives
    neg = torch.nn.functional.normalize(q + spread * mk(b, nn, d), dim=-1)
    other = torch.nn.functional.normalize(neg[:, :1] + 0.5 * mk(b, 1, d), dim=-1)     # close to a negative: the second term is active
    return q, pos, neg, other


def cloud_inputs(seed=3, n=777):
    rs = np.random.RandomState(seed)
    return rs.standard_normal((n, 3)) * np.array([20.0, 12.0, 3.0]) + np.array([5.7e6, 6.2e5, 110.0])


def call_loss(fn, name, kw, q, pos, neg, other, m1=0.5, m2=0.2):
    q, pos, neg, other = [t.clone().requires_grad_(True) for t in (q, pos, neg, other)]
    if name == "triplet_loss":
        v = fn(q, pos, neg, m1, **kw)
    else:
        v = fn(q, pos, neg, other, m1, m2, **kw)
    v.backward()
    grads = [t.grad if t.grad is not None else torch.zeros_like(t) for t in (q, pos, neg, other)]
    return v.detach().numpy(), [g.numpy() for g in grads]


def main():
    sys.dont_write_bytecode = True
    for name in ("chamfer", "emd"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.path.insert(0, REF)
    from losses import pointnetvlad_loss as ref
    from utils import loading_pointclouds as lp
    out = {}
    q, pos, neg, other = loss_inputs()
    for i, (name, kw) in enumerate(LOSS_CASES):
        v, grads = call_loss(getattr(ref, name), name, kw, q, pos, neg, other)
        out[f"loss{i}_value"] = v
        for t, g in zip("qpno", grads):
            out[f"loss{i}_grad_{t}"] = g.astype(np.float32)
        print(name, kw, float(v))
    # patch-feature contrastive loss on lists of vectors (pointnetvlad_loss.py:169-186)
    g = torch.Generator().manual_seed(11)
    lists = [[torch.randn(256, generator=g, dtype=torch.float64) for _ in range(9)] for _ in range(3)]
    out["contrastive_value"] = ref.contrastive_loss(lists[0], lists[1], lists[2], 0.5).numpy()
    out["contrastive_value_nopos"] = ref.contrastive_loss(lists[0], [], lists[2], 25.0).numpy()
    # loader + normalisation (loading_pointclouds.py:14-63)
    pc = cloud_inputs()
    path = "/tmp/_pa_golden_cloud.bin"
    pc.astype(np.float64).tofile(path)
    loaded = lp.load_pc_file(path)
    assert np.array_equal(loaded, pc)
    n1, meta = lp.normalize_point_cloud(loaded.copy(), return_norm_meta=True)
    n2 = lp.normalize_point_cloud(loaded.copy(), zoom=False)
    out["norm_zoom"] = n1
    out["norm_scale"] = np.array(meta["scale"])
    out["norm_trans"] = meta["trans"]
    out["norm_nozoom"] = n2
    os.remove(path)
    np.savez_compressed(os.path.join(GOLD, "losses.npz"), **out)
    print("wrote losses.npz", os.path.getsize(os.path.join(GOLD, "losses.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
