"""Generate tests/golden/patch_pairs.npz by EXECUTING the reference's own patch-pair loop (build container only).

The selection code is not a function in the reference: it is the body of ``if use_patch_feature_contrast:`` inside ``train_one_epoch``
(place_recognition/train_place_recognition.py:308-385).  This script reads those lines from /root/reference at generation time, dedents
them and executes them with the local variables the loop expects (nothing of the reference's text is stored); the overlap tables are
real ``Uint32Pair`` protobuf messages of the reference's ``datasets/query_pos_neg_dataset_pb2``.  Two passes with identical seeds:
pass A with index-coded features (column 0 = row number) recovers the index lists from what the loop hands to the loss function, pass B
with real features gives the reference's ``contrastive_loss`` values.  The oracle restatement (oracle/patch_pairs_cpu.py) is checked
against pass A before anything is written.

Usage: python -m oracle.gen_pairs_golden
"""
import os
import random
import sys
import textwrap
import types

import numpy as np
import torch

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
NPOINTS, M0, MARGIN = 4096, 1024, 0.5
CASES = [  # name, records per pair, epoch (> 10 and hard mining => hard-only), use_hard_negative_patch_mining, seed
    ("soft", (120, 700), 3, False, 11),
    ("hard", (90, 640), 12, True, 12),
]


def reference_loop_source():
    lines = open(os.path.join(REF, "place_recognition", "train_place_recognition.py")).read().splitlines()
    beg = next(i for i, l in enumerate(lines) if l.strip() == "if use_patch_feature_contrast:")
    end = next(i for i in range(beg, len(lines)) if lines[i].strip() == "loss_sum = 0.0")
    return textwrap.dedent("\n".join(lines[beg:end]))


def make_case(nrec_pair, seed):
    rs = np.random.RandomState(seed)
    centers = [rs.permutation(NPOINTS)[:M0].astype(np.int32) for _ in range(3)]        # clouds 0 (query), 1, 2 (positives)
    tables = {}
    for n, nrec in zip((1, 2), nrec_pair):
        recs = []
        for _ in range(nrec):
            idx1 = int(centers[0][rs.randint(M0)]) if rs.rand() < 0.7 else int(rs.randint(NPOINTS))

            def lst(maxlen, p_center):
                ln = rs.randint(0, maxlen + 1)
                return [int(centers[n][rs.randint(M0)]) if rs.rand() < p_center else int(rs.randint(NPOINTS)) for _ in range(ln)]
            recs.append({"idx1": idx1, "near_indices2": lst(12, 0.5), "far_indices2": lst(7, 0.6), "bad_far_indices2": lst(5, 0.6)})
        tables[(0, n)] = recs
    return centers, tables


def run_reference(src, centers, tables, feats, epoch, hard, seed, pb):
    calls = []

    def recorder(q, p, n, margin):
        calls.append((torch.stack(q), torch.stack(p), torch.stack(n), margin))
        return ref_contrastive(q, p, n, margin)
    nn_dict = {key: [pb.Uint32Pair(idx1=r["idx1"], near_indices2=r["near_indices2"], far_indices2=r["far_indices2"], bad_far_indices2=r["bad_far_indices2"])
                     for r in recs] for key, recs in tables.items()}
    ns = dict(use_patch_feature_contrast=True, cur_loss={}, nn_dict=nn_dict, cloud_indices=[0, 1, 2],
              center_indices=[torch.from_numpy(c).view(1, -1) for c in centers], patch_features=feats, epoch=epoch, hard_neg_epoch_for_patch_align=10,
              use_hard_negative_patch_mining=hard, device="cpu", loss_func={"patch_recon_a2b": recorder}, args={"MARGIN_1": MARGIN},
              num_iter_loss={"patch_recon_a2b": 0}, np=np, torch=torch, random=random)
    random.seed(seed)
    np.random.seed(seed)
    exec(compile(src, "<reference train_one_epoch lines>", "exec"), ns)
    return calls, ns["cur_loss"]["patch_recon_a2b"]


def main():
    global ref_contrastive
    sys.dont_write_bytecode = True
    for name in ("chamfer", "emd"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.path.insert(0, REF)
    from datasets import query_pos_neg_dataset_pb2 as pb
    from losses import pointnetvlad_loss as ref_losses
    ref_contrastive = ref_losses.contrastive_loss
    from oracle import patch_pairs_cpu
    src = reference_loop_source()
    out = {"npoints": NPOINTS, "m0": M0, "margin": MARGIN}
    for name, nrec_pair, epoch, hard, seed in CASES:
        centers, tables = make_case(nrec_pair, seed)
        hard_only = epoch > 10 and hard
        coded = [torch.zeros(M0, 4, dtype=torch.float64) for _ in range(3)]
        for c, t in enumerate(coded):
            t[:, 0] = torch.arange(M0, dtype=torch.float64)
            t[:, 1] = c
        g = torch.Generator().manual_seed(seed)
        feats = [torch.nn.functional.normalize(torch.randn(M0, 256, generator=g, dtype=torch.float64)) for _ in range(3)]
        calls_a, _ = run_reference(src, centers, tables, coded, epoch, hard, seed, pb)
        calls_b, total = run_reference(src, centers, tables, feats, epoch, hard, seed, pb)
        assert len(calls_a) == len(calls_b) == 2, "both pairs must produce triplets in the fixture"
        # the oracle restatement under the same seeds, pair after pair, must give the same lists
        random.seed(seed)
        np.random.seed(seed)
        for j, (key, recs) in enumerate(tables.items()):
            q, p, n, _ = calls_a[j]
            ref = [q[:, 0].numpy().astype(np.int64), p[:, 0].numpy().astype(np.int64), n[:, 0].numpy().astype(np.int64)]
            assert (q[:, 1] == 0).all() and (p[:, 1] == key[1]).all() and (n[:, 1] == key[1]).all()
            (i1, p2, n2), kept = patch_pairs_cpu.select_pairs(centers[0], centers[key[1]], recs, hard_only, random, np.random)
            assert i1 == ref[0].tolist() and p2 == ref[1].tolist() and n2 == ref[2].tolist(), f"oracle restatement differs from the reference loop ({name}, {key})"
            out[f"{name}/pair{j}/indices1"], out[f"{name}/pair{j}/pos_indices2"], out[f"{name}/pair{j}/neg_indices2"] = ref
            out[f"{name}/pair{j}/loss"] = calls_b[j][3] * 0 + float(ref_contrastive(list(calls_b[j][0]), list(calls_b[j][1]), list(calls_b[j][2]), MARGIN))
            out[f"{name}/pair{j}/kept_records"] = np.array([k for k, *_ in kept], np.int64)
        out[f"{name}/loss_total"] = float(total)
        out[f"{name}/seed"], out[f"{name}/epoch"], out[f"{name}/hard"] = seed, epoch, int(hard)
        out[f"{name}/nrec"] = np.array(nrec_pair)
        print(name, "triplets per pair:", [len(calls_a[j][0]) for j in range(2)], "loss", float(total))
    np.savez_compressed(os.path.join(GOLD, "patch_pairs.npz"), **out)


if __name__ == "__main__":
    main()
