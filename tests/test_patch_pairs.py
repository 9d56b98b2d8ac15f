"""Patch overlap-pair selection + contrastive patch-feature term (train_place_recognition.py:308-385).

tests/golden/patch_pairs.npz holds what the REFERENCE's own loop produced (oracle/gen_pairs_golden.py executes those lines of
train_one_epoch on synthetic Uint32Pair tables): index lists per cloud pair and contrastive_loss values, soft and hard-negative mode,
tables below and above the 500-record cap.

CPU: the oracle restatement and the CSR packing against the fixture.  GPU: csrc/patch_pairs.hip against the fixture -- query and
positive positions identical to the reference's lists; the negative draw is a different random stream by design, so every drawn
negative is checked to be one of that record's far positions and the draw to be spread over them; the loss of the reference's exact
triplets equals the reference's value.
"""
import random

import numpy as np
import pytest
import torch

from tests._util import golden


def _cases():
    from oracle.gen_pairs_golden import CASES, make_case
    for name, nrec_pair, epoch, hard, seed in CASES:
        centers, tables = make_case(nrec_pair, seed)
        yield name, centers, tables, (epoch > 10 and hard), seed


def test_oracle_restatement_reproduces_the_reference_loop():
    from oracle import patch_pairs_cpu
    z = golden("patch_pairs")
    for name, centers, tables, hard_only, seed in _cases():
        random.seed(seed)
        np.random.seed(seed)
        for j, (key, recs) in enumerate(tables.items()):
            (i1, p2, n2), kept = patch_pairs_cpu.select_pairs(centers[0], centers[key[1]], recs, hard_only, random, np.random)
            assert i1 == z[f"{name}/pair{j}/indices1"].tolist()
            assert p2 == z[f"{name}/pair{j}/pos_indices2"].tolist()
            assert n2 == z[f"{name}/pair{j}/neg_indices2"].tolist()
            assert [k for k, *_ in kept] == z[f"{name}/pair{j}/kept_records"].tolist()


def test_csr_packing_and_far_candidate_rule():
    from oracle import patch_pairs_cpu
    from patchaugnet_amd.patch_pairs import OverlapTable
    for name, centers, tables, hard_only, seed in _cases():
        for key, recs in tables.items():
            t = OverlapTable.from_records(recs)
            assert len(t) == len(recs)
            for hard in (False, True):
                off, vals = t.far_candidates(hard)
                for k in (0, 1, len(recs) // 2, len(recs) - 1):
                    want = patch_pairs_cpu.far_candidates(recs[k], hard)
                    want = want if isinstance(want, list) else [want]
                    assert vals[off[k]:off[k + 1]].tolist() == want
            ks = [5, 0, len(recs) - 1, 5]
            idx1, no, nv, fo, fv = t.take(ks, hard_only)
            for i, k in enumerate(ks):
                assert idx1[i] == recs[k]["idx1"]
                assert nv[no[i]:no[i + 1]].tolist() == recs[k]["near_indices2"]
    empty = OverlapTable.from_records([{"idx1": 1, "near_indices2": [], "far_indices2": [], "bad_far_indices2": []}])
    off, vals = empty.far_candidates(False)
    assert off.tolist() == [0, 0] and len(vals) == 0


@pytest.mark.gpu
def test_device_selection_matches_the_reference_loop():
    from oracle import patch_pairs_cpu
    from patchaugnet_amd import patch_pairs
    z = golden("patch_pairs")
    for name, centers, tables, hard_only, seed in _cases():
        random.seed(seed)
        np.random.seed(seed)
        for j, (key, recs) in enumerate(tables.items()):
            cm, cn = torch.from_numpy(centers[0]).cuda(), torch.from_numpy(centers[key[1]]).cuda()
            state = random.getstate()
            sel = patch_pairs.select_patch_pairs(cm, cn, recs, int(z["npoints"]), hard_only=hard_only, seed=seed + j, py_random=random)
            i1, p2, n2 = sel.trimmed()
            assert i1.tolist() == z[f"{name}/pair{j}/indices1"].tolist(), (name, j)
            assert p2.tolist() == z[f"{name}/pair{j}/pos_indices2"].tolist(), (name, j)
            # negatives: one of the record's far positions each; over the whole pair the draw is not stuck on the first candidate
            random.setstate(state)
            _, kept = patch_pairs_cpu.select_pairs(centers[0], centers[key[1]], recs, hard_only, random, np.random.RandomState(0))
            at, multi, first = 0, 0, 0
            for k, q, pos, negs in kept:
                got = n2[at:at + len(pos)]
                assert set(got.tolist()) <= set(negs), (name, j, k)
                if len(negs) > 1:
                    multi += len(pos)
                    first += int((got == negs[0]).sum())
                at += len(pos)
            assert at == len(n2)
            if multi >= 40:
                assert 0.15 * multi < first < 0.85 * multi, (first, multi)
            # same seed, same draw; another seed, another draw
            random.setstate(state)
            again = patch_pairs.select_patch_pairs(cm, cn, recs, int(z["npoints"]), hard_only=hard_only, seed=seed + j, py_random=random).trimmed()
            assert again[2].tolist() == n2.tolist()


@pytest.mark.gpu
def test_contrastive_term_on_the_reference_triplets():
    from patchaugnet_amd import patch_pairs
    z = golden("patch_pairs")
    for name, centers, tables, hard_only, seed in _cases():
        g = torch.Generator().manual_seed(seed)
        feats = [torch.nn.functional.normalize(torch.randn(int(z["m0"]), 256, generator=g, dtype=torch.float64)).cuda() for _ in range(3)]
        vals = []
        for j, key in enumerate(tables):
            i1, p2, n2 = (torch.from_numpy(z[f"{name}/pair{j}/{k}"]).int().cuda() for k in ("indices1", "pos_indices2", "neg_indices2"))
            pad = 17                                                          # padded like select_patch_pairs pads: the mask must hide the tail
            cat = lambda t: torch.cat([t, torch.zeros(pad, dtype=torch.int32, device="cuda")])
            sel = patch_pairs.PairSelection(cat(i1), cat(p2), cat(n2), torch.tensor(len(i1), device="cuda"), len(i1) + pad)
            loss, has = patch_pairs.contrastive_loss_selected(feats[0], feats[key[1]], sel, float(z["margin"]))
            assert has.item() == 1.0
            assert abs(loss.item() - float(z[f"{name}/pair{j}/loss"])) <= 1e-12
            vals.append(loss.item())
        assert abs(np.mean(vals) - float(z[f"{name}/loss_total"])) <= 1e-12


@pytest.mark.gpu
def test_training_step_with_the_patch_contrast_term():
    """One reduced-size step through train.training_step with the overlap tables: the term is finite, positive, and has a gradient path
    into the backbone (patch features come from the FP levels)."""
    from patchaugnet_amd import configs, patch_aug_net
    from patchaugnet_amd.train import training_step
    from patchaugnet_amd.weights import seeded_state_dict
    n = 1024
    cfg = configs.scaled_config(configs.patch_aug_net_config(), n)
    m = patch_aug_net.Network(param=cfg, use_a2a_recon=True, use_l2_norm=True)
    m.load_state_dict(seeded_state_dict(m.state_dict()))
    m = m.cuda()
    g = torch.Generator().manual_seed(2)
    q, pos, neg, oth = (torch.rand(1, k, n, 3, generator=g) * 2 - 1 for k in (1, 2, 4, 1))
    rs = np.random.RandomState(4)
    recs = lambda: [{"idx1": int(rs.randint(n)), "near_indices2": rs.randint(0, n, 40).tolist(), "far_indices2": rs.randint(0, n, 30).tolist(),
                     "bad_far_indices2": rs.randint(0, n, 10).tolist()} for _ in range(300)]
    nn_dict = {(0, 1): recs(), (0, 2): recs()}
    opt = torch.optim.Adam(m.parameters(), lr=1e-5)
    args = dict(__import__("patchaugnet_amd.train", fromlist=["DEFAULTS"]).DEFAULTS, TRAIN_NEGATIVES_PER_QUERY=4)
    out = training_step(m, opt, q, pos, neg, oth, nn_dict=nn_dict, num_points=n, args=args, use_patch_feature_contrast=True)
    assert "patch_recon_a2b" in out and np.isfinite(out["patch_recon_a2b"]) and out["patch_recon_a2b"] > 0
    assert np.isfinite(out["total"])
