"""Generate tests/golden/recall.npz by running the REFERENCE's own SceneDataSet.get_recall_precision (build container only).

SURVEY.md appendix D5: ``open3d`` (absent here, used only by get_fpfh) is stubbed, ``datasets.scene_dataset`` is imported from
/root/reference unmodified, a SceneDataSet is filled in memory with a synthetic multi-trip route (records, sizes, global
descriptors, positive tuples) so that its ``load`` finds nothing to read, and its ``get_recall_precision`` is driven by
the loop of datasets/place_recognition_dataset.py:52-70.  Only inputs and results are written to the repo.

Usage: python -m oracle.gen_recall_golden
"""
import os
import sys
import types

import numpy as np

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def synthetic_route(seed, sizes, dim=256, noise=1.6, pos_radius=25.0):
    """Trips along one shared path; descriptor = smooth random features of the position + per-submap noise."""
    rng = np.random.default_rng(seed)
    length = 1500.0
    freq = rng.normal(size=(dim, 2)) / 60.0
    phase = rng.uniform(0, 2 * np.pi, size=dim)
    xy, desc = [], []
    for n in sizes:
        s = np.sort(rng.uniform(0, length, size=n))
        p = np.stack([s, 40.0 * np.sin(s / 150.0)], axis=1) + rng.normal(scale=1.5, size=(n, 2))
        d = np.cos(p @ freq.T + phase) + noise * rng.normal(size=(n, dim))
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        xy.append(p)
        desc.append(d.astype(np.float32))
    xy, desc = np.concatenate(xy), np.concatenate(desc)
    starts = np.concatenate([[0], np.cumsum(sizes)])
    tuples = {}
    for q in range(len(sizes)):
        for r in range(len(sizes)):
            t = {}
            for i in range(starts[q], starts[q + 1]):
                dd = np.linalg.norm(xy[starts[r]:starts[r + 1]] - xy[i], axis=1)
                pos = [int(starts[r] + j) for j in np.nonzero(dd < pos_radius)[0] if starts[r] + j != i]
                t[int(i)] = pos
            tuples[q, r] = t
    return xy, desc, tuples


def run_reference(sizes, xy, desc, tuples, top_k, skip_trip_itself):
    sys.dont_write_bytecode = True
    sys.modules.setdefault("open3d", types.ModuleType("open3d"))
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import pandas as pd
    from sklearn.neighbors import KDTree
    from datasets.scene_dataset import QueryPosNegTuple, SceneDataSet
    ds = SceneDataSet("oxford", for_training=False)
    ds.trip_names = [f"trip{i}" for i in range(len(sizes))]
    ds.records = pd.DataFrame({"file": [f"f{i}.bin" for i in range(len(xy))], "northing": xy[:, 0], "easting": xy[:, 1]})
    ds.records_size_list = list(sizes)
    ds.global_desc_list = desc
    for key, t in tuples.items():
        d = {}
        for i, pos in t.items():
            tp = QueryPosNegTuple()
            tp.positive_indices = list(pos)
            d[i] = tp
        ds.query_pos_neg_tuples_dict[key] = d
    out = {}
    sample_indices = ds.get_indices_in_dataset()
    for r in range(len(sizes)):                                           # place_recognition_dataset.py:52-70
        database_indices = sample_indices[r]
        tree = KDTree(ds.global_desc_list[database_indices])
        for q in range(len(sizes)):
            if skip_trip_itself and q == r:
                continue
            out[q, r] = ds.get_recall_precision(tree, database_indices, q, r, top_k=top_k, skip_trip_itself=skip_trip_itself)
    return out


def main():
    os.makedirs(GOLD, exist_ok=True)
    cases = {"a": dict(seed=11, sizes=[130, 120, 141, 110], top_k=25, skip=False),
             "b": dict(seed=12, sizes=[650, 250, 350], top_k=5, skip=False),     # thresholds 6 (6.5 -> even) > top_k, 2 (2.5 -> even), 4 (3.5 -> even)
             "c": dict(seed=13, sizes=[60, 75, 50], top_k=25, skip=True)}
    blob = {}
    for tag, c in cases.items():
        xy, desc, tuples = synthetic_route(c["seed"], c["sizes"])
        res = run_reference(c["sizes"], xy, desc, tuples, c["top_k"], c["skip"])
        blob[f"{tag}_sizes"] = np.array(c["sizes"])
        blob[f"{tag}_top_k"] = np.array(c["top_k"])
        blob[f"{tag}_skip"] = np.array(c["skip"])
        blob[f"{tag}_seed"] = np.array(c["seed"])                       # inputs are re-made from the seed (synthetic_route); checksums pin them
        blob[f"{tag}_desc_sum"] = np.array([desc.astype(np.float64).sum(), np.abs(desc.astype(np.float64)).sum()])
        blob[f"{tag}_desc_head"] = desc[:4]
        keys = sorted(res)
        blob[f"{tag}_pairs"] = np.array(keys)
        blob[f"{tag}_recall"] = np.stack([res[k][0] for k in keys])
        blob[f"{tag}_precision"] = np.stack([res[k][1] for k in keys])
        blob[f"{tag}_opr"] = np.array([res[k][2] for k in keys])
        blob[f"{tag}_lost"] = np.array([res[k][3] for k in keys])
        blob[f"{tag}_threshold"] = np.array([res[k][4] for k in keys])
        blob[f"{tag}_states"] = np.concatenate([[qr["state"] for qr in res[k][5]] for k in keys]).astype(np.int8)
        blob[f"{tag}_nstates"] = np.array([len(res[k][5]) for k in keys])
        blob[f"{tag}_num_eval"] = np.array([res[k][6] for k in keys])
        blob[f"{tag}_num_ref"] = np.array([res[k][7] for k in keys])
        print(tag, "pairs", len(keys), "recall@1 mean %.2f" % np.mean([res[k][0][0] for k in keys if k[0] != k[1]]),
              "thresholds", sorted(set(blob[f"{tag}_threshold"].tolist())))
    np.savez_compressed(os.path.join(GOLD, "recall.npz"), **blob)
    print("wrote", os.path.join(GOLD, "recall.npz"), os.path.getsize(os.path.join(GOLD, "recall.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
