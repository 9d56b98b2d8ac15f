"""Deterministic, key-seeded weights (there are no checkpoints in the image).

``seeded_state_dict(template)`` fills every entry of a state-dict from a
generator seeded by the entry's NAME, so the reference model (in the build
container), the CPU oracle and the HIP engine (on the GPU box) all get the same
numbers without shipping a checkpoint.  BatchNorm running statistics are made
non-trivial on purpose so that BN folding is actually exercised.
"""
import zlib
from collections import OrderedDict

import numpy as np
import torch


def _fill(key, shape, dtype, seed):
    rs = np.random.RandomState((zlib.crc32(key.encode()) ^ seed) & 0x7FFFFFFF)
    leaf = key.rsplit(".", 1)[-1]
    if leaf == "num_batches_tracked":
        return torch.tensor(100, dtype=dtype)
    n = int(np.prod(shape)) if len(shape) else 1
    if leaf == "running_var":
        a = rs.uniform(0.5, 1.5, n)
    elif leaf == "running_mean":
        a = 0.1 * rs.standard_normal(n)
    elif len(shape) <= 1:
        a = rs.uniform(0.5, 1.5, n) if leaf == "weight" else 0.1 * rs.standard_normal(n)
    elif "cluster_weights" in leaf:
        a = rs.standard_normal(n) / np.sqrt(shape[-2])
    elif "hidden" in leaf or "gating_weights" in leaf:
        a = rs.standard_normal(n) / np.sqrt(shape[-1])
    else:  # conv / linear weight (out, in, ...)
        fan_in = int(np.prod(shape[1:]))
        a = rs.standard_normal(n) * np.sqrt(2.0 / fan_in)
    return torch.from_numpy(np.asarray(a, dtype=np.float32).reshape(shape)).to(dtype)


def seeded_state_dict(template, seed=1234):
    """template: a state_dict (or {name: tensor}) giving names, shapes, dtypes."""
    out = OrderedDict()
    for k, v in template.items():
        out[k] = _fill(k, tuple(v.shape), v.dtype, seed)
    return out


def synthetic_submaps(batch, num_points=4096, seed=1234, kind="uniform"):
    """Synthetic 4096-point submaps, pre-normalised to about [-1, 1].

    kind "uniform": torch.rand(B,1,N,3, seed)*2-1.
    kind "street" : points on a few random planes plus 5 % exact duplicates (exercises ties).
    Returns float32 (B, 1, N, 3) on CPU.
    """
    if kind == "uniform":
        g = torch.Generator().manual_seed(seed)
        return torch.rand(batch, 1, num_points, 3, generator=g) * 2 - 1
    rs = np.random.RandomState(seed)
    out = np.empty((batch, num_points, 3), dtype=np.float32)
    for b in range(batch):
        n_planes = rs.randint(3, 6)
        per = num_points // n_planes
        pts = []
        for p in range(n_planes):
            cnt = per if p < n_planes - 1 else num_points - per * (n_planes - 1)
            origin = rs.uniform(-0.5, 0.5, 3)
            u, v = rs.standard_normal(3), rs.standard_normal(3)
            u /= np.linalg.norm(u)
            v -= u * (u @ v)
            v /= np.linalg.norm(v)
            st = rs.uniform(-0.7, 0.7, (cnt, 2))
            pts.append(origin + st[:, :1] * u + st[:, 1:] * v + 0.002 * rs.standard_normal((cnt, 3)))
        pts = np.concatenate(pts).astype(np.float32)
        dup = rs.choice(num_points, num_points // 20, replace=False)
        src = rs.choice(num_points, num_points // 20, replace=False)
        pts[dup] = pts[src]
        pts = pts[rs.permutation(num_points)]
        out[b] = np.clip(pts, -1, 1)
    return torch.from_numpy(out).unsqueeze(1)
