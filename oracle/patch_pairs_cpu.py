"""CPU restatement of the patch overlap-pair selection (TEST INFRASTRUCTURE: only tests/ may import this).

Follows the loops of ``train_one_epoch``, place_recognition/train_place_recognition.py:312-372, statement by statement, including the
host random streams (``random.sample`` for tables above 500 records, ``np.random.choice`` for the negative draw) so that, seeded like the
reference, it reproduces the reference's index lists exactly (pinned by tests/golden/patch_pairs.npz, which oracle/gen_pairs_golden.py
produces by EXECUTING those reference lines).
"""
import numpy as np


def far_candidates(rec, hard_only):
    """:344-355 -- hard mode: bad_far_indices2; else the quirk: the loop keeps ONE element, the last even-indexed one of far + bad_far."""
    if hard_only:
        return list(rec["bad_far_indices2"])
    temp = list(rec["far_indices2"]) + list(rec["bad_far_indices2"])
    out = []
    for far_i in range(0, len(temp), 2):
        out = temp[far_i]
    return out


def select_pairs(m_center, n_center, records, hard_only, py_random, np_random):
    """records: list of dicts idx1 / near_indices2 / far_indices2 / bad_far_indices2.  Returns (indices1, pos_indices2, neg_indices2)
    and, per kept record, (record index, query position, positive positions, negative candidate positions)."""
    m_center, n_center = np.asarray(m_center), np.asarray(n_center)
    indices1, pos_indices2, neg_indices2, kept = [], [], [], []
    k_list = [n for n in range(len(records))]
    if len(k_list) > 500:                                                   # :330-331
        k_list = py_random.sample(k_list, 500)
    for k in k_list:
        rec = records[k]
        idx1 = np.where(m_center == rec["idx1"])[0].tolist()                # :333
        if len(idx1) == 0:
            continue
        pos_idx2 = np.where(np.isin(n_center, list(rec["near_indices2"])))[0].tolist()     # :339-341
        if len(pos_idx2) == 0:
            continue
        neg_idx2 = np.where(np.isin(n_center, far_candidates(rec, hard_only)))[0].tolist()  # :356
        if len(neg_idx2) == 0:
            continue
        kept.append((k, idx1[0], list(pos_idx2), list(neg_idx2)))
        idx1 = (np.ones(len(pos_idx2), dtype="int32") * idx1[0]).tolist()
        neg_idx2 = np_random.choice(neg_idx2, len(pos_idx2), replace=True).tolist()          # :360
        indices1 += idx1
        pos_indices2 += pos_idx2
        neg_indices2 += neg_idx2
    return (indices1, pos_indices2, neg_indices2), kept
