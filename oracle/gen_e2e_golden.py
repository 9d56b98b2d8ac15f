"""End-to-end Recall fixture: point clouds -> descriptors -> retrieval (north_star "Recall@1 within
0.1 % of the reference").  Build container only; writes tests/golden/e2e_recall.npz.

Chain pinned here, every link executed by the REFERENCE's own Python where it can run:
  clouds (seeded synthetic multi-trip drive past 36 places of distinct global shape, this file)
    -> descriptors: the CPU oracle model (oracle/models_cpu.py, itself pinned to the reference's classes by gen_golden.py)
    -> Recall@N / precision / top-1 %: the reference's SceneDataSet.get_recall_precision driven like
       datasets/place_recognition_dataset.py:52-70 (oracle/gen_recall_golden.run_reference), i.e. evaluate.py:167-237's numbers.
The -m gpu test (tests/test_gpu_e2e_recall.py) re-makes the clouds from the seed, extracts descriptors with the HIP engine through
distributed.extract_dataset, retrieves with the HIP kNN and must land within 0.1 percentage points of these numbers.

Usage: python -m oracle.gen_e2e_golden
"""
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")

SIZES = [34, 30, 32, 28]          # four trips, 124 submaps
SEED = 77
NUM_POINTS = 4096
PLACES = 36                       # distinct places; every trip passes a random subset of them
POS_RADIUS = 8.0                  # metres: true positives of a query = submaps of the reference trip closer than this
ROUTE = 700.0
FRESH = 0.30                      # share of a visit's returns that are re-drawn (transients, occlusion) instead of the place's fixed returns


# The Oxford-sized set of SURVEY.md section 8(d) config 3: 23 trips (datasets/dataset_info.py:127-132) of ~130 submaps each (the real count
# is not in the repository), one shared road, positives closer than 25 m.
OX_SIZES = [132, 127, 135, 129, 131, 126, 134, 130, 128, 133, 125, 136, 130, 129, 131, 127, 134, 132, 128, 130, 126, 135, 131]      # 2 999 submaps
OX_SEED = 1234
OX_PLACES = 160
OX_ROUTE = 3200.0                 # 20 m between places
OX_POS_RADIUS = 25.0


def trip_stops(seed=SEED, sizes=SIZES, places=PLACES):
    """Which place every submap of every trip was taken at (sorted along the road within a trip)."""
    rng = np.random.default_rng(seed + 1)
    return [np.sort(rng.choice(places, n, replace=False)) for n in sizes]


def trip_positions(seed=SEED, sizes=SIZES, places=PLACES, route=ROUTE):
    """(sum sizes, 2) northing / easting: the place's position on the road plus ~1.5 m of per-visit offset."""
    rng = np.random.default_rng(seed + 2)
    stops = np.linspace(15.0, route - 15.0, places)
    return np.concatenate([np.stack([stops[st] + rng.normal(scale=1.5, size=len(st)), rng.normal(scale=0.6, size=len(st))], 1)
                           for st in trip_stops(seed, sizes, places)])


def submap(place_seed, visit_seed, num_points=NUM_POINTS, fresh=FRESH):
    """(num_points, 3) fp32 cloud in about [-1, 1].  A place is a mixture of 6-40 anisotropic blobs with its own extent and heading
    (drawn from place_seed, together with the fixed surface points the sensor returns there); a visit re-draws a share `fresh` of the
    returns, adds 2 mm-scale range noise and a small pose offset, and shuffles the point order (all from visit_seed).  Random-init
    descriptors separate places by this global shape only moderately, so Recall@1 sits in its sensitive mid range."""
    pr = np.random.default_rng(place_seed)
    nb = int(pr.integers(6, 40))
    ext = 0.25 + 0.75 * pr.random(3)
    c = (pr.random((nb, 3)) * 2 - 1) * ext
    sp = 0.02 + 0.2 * pr.random((nb, 3)) * ext
    w = pr.random(nb) + 0.2
    w /= w.sum()
    th = pr.uniform(0, np.pi)
    which = pr.choice(nb, size=num_points, p=w)
    pts = c[which] + sp[which] * pr.normal(size=(num_points, 3))
    vr = np.random.default_rng(visit_seed)
    redo = vr.random(num_points) < fresh
    wh2 = vr.choice(nb, size=num_points, p=w)
    p2 = c[wh2] + sp[wh2] * vr.normal(size=(num_points, 3))
    pts[redo] = p2[redo]
    pts = pts + 0.002 * vr.normal(size=pts.shape)
    pts = pts[vr.permutation(num_points)]
    rot = np.array([[np.cos(th), -np.sin(th), 0.0], [np.sin(th), np.cos(th), 0.0], [0.0, 0.0, 1.0]])
    pts = pts @ rot.T + vr.normal(scale=0.01, size=3)
    return np.clip(pts, -1.0, 1.0).astype(np.float32)


def clouds(lo, hi, seed=SEED, sizes=SIZES, num_points=NUM_POINTS, places=PLACES):
    """Records lo..hi-1 as a (hi - lo, 1, N, 3) fp32 CPU tensor (the loader callback of distributed.extract_dataset)."""
    st = np.concatenate(trip_stops(seed, sizes, places))
    return torch.from_numpy(np.stack([submap(seed * 100003 + int(st[i]), seed * 7919 + 1000 + i, num_points) for i in range(lo, hi)])).unsqueeze(1)


def positives(xy, sizes=SIZES, radius=POS_RADIUS):
    starts = np.concatenate([[0], np.cumsum(sizes)])
    tuples = {}
    for q in range(len(sizes)):
        for r in range(len(sizes)):
            t = {}
            for i in range(starts[q], starts[q + 1]):
                d = np.linalg.norm(xy[starts[r]:starts[r + 1]] - xy[i], axis=1)
                t[int(i)] = [int(starts[r] + j) for j in np.nonzero(d < radius)[0] if starts[r] + j != i]
            tuples[q, r] = t
    return tuples


def oracle_descriptors(x):
    from oracle import models_cpu
    from patchaugnet_amd import configs, patch_aug_net
    from patchaugnet_amd.weights import seeded_state_dict
    cfg = configs.patch_aug_net_config()
    sd = seeded_state_dict(patch_aug_net.Network(param=cfg, use_a2a_recon=True, use_l2_norm=True).state_dict())
    out = []
    with torch.no_grad():
        for lo in range(0, x.shape[0], 8):
            d, _, _ = models_cpu.patch_aug_net_forward(sd, cfg, x[lo:lo + 8])
            out.append(d)
    return torch.cat(out).numpy()


def main():
    from oracle.gen_recall_golden import run_reference
    n = sum(SIZES)
    x = clouds(0, n)
    xy = trip_positions()
    tuples = positives(xy)
    desc = oracle_descriptors(x)
    res = run_reference(SIZES, xy, desc, tuples, top_k=25, skip_trip_itself=True)
    keys = sorted(res)
    blob = {"sizes": np.array(SIZES), "seed": np.array(SEED), "top_k": np.array(25),
            "cloud_checksum": np.array([x.double().sum().item(), x.double().abs().sum().item()]), "cloud_head": x[0, 0, :8].numpy(),
            "oracle_desc": desc.astype(np.float32),
            "pairs": np.array(keys), "recall": np.stack([res[k][0] for k in keys]), "precision": np.stack([res[k][1] for k in keys]),
            "opr": np.array([res[k][2] for k in keys]), "num_eval": np.array([res[k][6] for k in keys])}
    rec = np.mean([res[k][0] for k in keys], 0)
    print("pairs", len(keys), "queries evaluated", blob["num_eval"].sum(), "recall@1 %.2f  @5 %.2f  top1%% %.2f" % (rec[0], rec[4], np.mean(blob["opr"])))
    np.savez_compressed(os.path.join(GOLD, "e2e_recall.npz"), **blob)
    print("wrote e2e_recall.npz", os.path.getsize(os.path.join(GOLD, "e2e_recall.npz")) // 1024, "KiB")


DESC_PROBES = 4


def desc_probe(desc):
    """(n, DESC_PROBES) float64 projections of the descriptors on fixed seeded unit directions: a few numbers per submap that pin the
    descriptors of a 3 000-submap set without committing 3 MB of them."""
    g = np.random.default_rng(4242)
    d = g.normal(size=(desc.shape[1], DESC_PROBES))
    d /= np.linalg.norm(d, axis=0, keepdims=True)
    return np.asarray(desc, dtype=np.float64) @ d


def main_oxford():
    """tests/golden/e2e_recall_oxford.npz: the same chain on the Oxford-sized set (23 trips, 2 999 submaps, 506 trip pairs)."""
    import time
    from oracle.gen_recall_golden import run_reference
    n = sum(OX_SIZES)
    t0 = time.time()
    desc = np.concatenate([oracle_descriptors(clouds(lo, min(lo + 64, n), OX_SEED, OX_SIZES, NUM_POINTS, OX_PLACES)) for lo in range(0, n, 64)])
    print("oracle descriptors of", n, "submaps in %.0f s" % (time.time() - t0))
    xy = trip_positions(OX_SEED, OX_SIZES, OX_PLACES, OX_ROUTE)
    tuples = positives(xy, OX_SIZES, OX_POS_RADIUS)
    res = run_reference(OX_SIZES, xy, desc, tuples, top_k=25, skip_trip_itself=True)
    keys = sorted(res)
    head = clouds(0, 1, OX_SEED, OX_SIZES, NUM_POINTS, OX_PLACES)
    blob = {"sizes": np.array(OX_SIZES), "seed": np.array(OX_SEED), "top_k": np.array(25), "cloud_head": head[0, 0, :8].numpy(),
            "desc_probe": desc_probe(desc),
            "pairs": np.array(keys), "recall": np.stack([res[k][0] for k in keys]).astype(np.float32),
            "precision": np.stack([res[k][1] for k in keys]).astype(np.float32),
            "opr": np.array([res[k][2] for k in keys]), "num_eval": np.array([res[k][6] for k in keys])}
    rec = np.mean([res[k][0] for k in keys], 0)
    print("pairs", len(keys), "queries evaluated", blob["num_eval"].sum(), "recall@1 %.2f  @5 %.2f  top1%% %.2f" % (rec[0], rec[4], np.mean(blob["opr"])))
    np.savez_compressed(os.path.join(GOLD, "e2e_recall_oxford.npz"), **blob)
    print("wrote e2e_recall_oxford.npz", os.path.getsize(os.path.join(GOLD, "e2e_recall_oxford.npz")) // 1024, "KiB")


if __name__ == "__main__":
    import sys
    main_oxford() if "oxford" in sys.argv[1:] else main()
