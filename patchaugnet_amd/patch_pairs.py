"""Patch overlap-pair selection + the contrastive patch-feature term of the training step, on the MI355X.

Reference: the Python loops in ``train_one_epoch`` (place_recognition/train_place_recognition.py:308-385).  For every (query cloud m,
positive cloud n) key of ``nn_dict`` the precomputed overlap table (datasets/scene_dataset.py:278-297: ``Uint32Pair`` protobuf records
with ``idx1 / near_indices2 / far_indices2 / bad_far_indices2`` in ORIGINAL point indices) is intersected with the two clouds' FPS centre
lists to give (query patch, positive patch, negative patch) triplets, whose features go into ``contrastive_loss``
(losses/pointnetvlad_loss.py:170-186).  The reference does this record by record in numpy with one H2D copy per selected index and
caps the table at 500 records per pair; here the table is packed once (CSR), two kernels (csrc/patch_pairs.hip) produce the triplets
on the device, and the loss is one gather + masked mean with no host synchronisation.

Kept exactly: the first-match lookup of ``idx1``, the ascending unique order of the positive positions, the three drop conditions, one
query copy per positive, ``random.sample(k_list, 500)`` on the host's ``random`` module for tables above 500 records, and the
far-candidate quirk of :345-355 (outside hard-negative mode the loop ``for far_i in range(0, len(t), 2): list_far = t[far_i]`` leaves ONE
candidate: the last even-indexed element of far + bad_far).  Different by design: the negative draw (``np.random.choice(neg, len(pos))``)
comes from a counter-based hash on the device -- same distribution, not numpy's stream.
"""
import random

import numpy as np
import torch
import torch.nn.functional as F

from ._lib import call, check_device, ptr


class OverlapTable:
    """CSR packing of one nn_dict value: a sequence of records with idx1 / near_indices2 / far_indices2 / bad_far_indices2."""

    def __init__(self, idx1, near_off, near, far_off, far, bad_off, bad):
        self.idx1, self.near_off, self.near = idx1, near_off, near
        self.far_off, self.far, self.bad_off, self.bad = far_off, far, bad_off, bad

    def __len__(self):
        return len(self.idx1)

    @staticmethod
    def _get(rec, name, pos):
        if isinstance(rec, dict):
            return rec[name]
        if isinstance(rec, (tuple, list)):
            return rec[pos]
        return getattr(rec, name)

    @classmethod
    def from_records(cls, records):
        idx1, lists = [], ([], [], [])
        for r in records:
            idx1.append(int(cls._get(r, "idx1", 0)))
            for j, name in enumerate(("near_indices2", "far_indices2", "bad_far_indices2")):
                lists[j].append(np.asarray(list(cls._get(r, name, j + 1)), dtype=np.int64))

        def csr(ls):
            off = np.zeros(len(ls) + 1, np.int64)
            if ls:
                off[1:] = np.cumsum([len(a) for a in ls])
            return off, (np.concatenate(ls) if ls and off[-1] else np.zeros(0, np.int64))
        (no, nv), (fo, fv), (bo, bv) = csr(lists[0]), csr(lists[1]), csr(lists[2])
        return cls(np.asarray(idx1, np.int64), no, nv, fo, fv, bo, bv)

    def far_candidates(self, hard_only):
        """(offsets, values) of the far candidates per record as the reference forms them (:345-355)."""
        if hard_only:
            return self.bad_off, self.bad
        nf, nb = np.diff(self.far_off), np.diff(self.bad_off)
        tot = nf + nb
        last = 2 * ((tot - 1) // 2)                         # last even index of far + bad_far (meaningless where tot == 0)
        has = tot > 0
        src_far = has & (last < nf)
        vals = np.zeros(len(tot), np.int64)
        vals[src_far] = self.far[(self.far_off[:-1] + last)[src_far]]
        src_bad = has & ~src_far
        vals[src_bad] = self.bad[(self.bad_off[:-1] + last - nf)[src_bad]]
        off = np.zeros(len(tot) + 1, np.int64)
        off[1:] = np.cumsum(has)
        return off, vals[has]

    def take(self, ks, hard_only):
        """Records ks (in that order) as int32 CSR arrays: idx1, near_off, near, far_off, far."""
        ks = np.asarray(ks, np.int64)

        def sub(off, vals):
            lens = (off[1:] - off[:-1])[ks]
            o = np.zeros(len(ks) + 1, np.int64)
            o[1:] = np.cumsum(lens)
            if o[-1] == 0:
                return o.astype(np.int32), np.zeros(1, np.int32)
            src = np.repeat(off[:-1][ks] - o[:-1], lens) + np.arange(o[-1])
            return o.astype(np.int32), vals[src].astype(np.int32)
        no, nv = sub(self.near_off, self.near)
        fo, fv = sub(*self.far_candidates(hard_only))
        return self.idx1[ks].astype(np.int32), no, nv, fo, fv


class PairSelection:
    """Triplets of one (m, n) cloud pair, padded to `cap` entries; `total` (0-dim int64 device tensor) of them are valid."""
    __slots__ = ("idx1", "pos2", "neg2", "total", "cap")

    def __init__(self, idx1, pos2, neg2, total, cap):
        self.idx1, self.pos2, self.neg2, self.total, self.cap = idx1, pos2, neg2, total, cap

    def trimmed(self):
        """(indices1, pos_indices2, neg_indices2) as the reference's Python lists hold them -- synchronises; for tests and inspection."""
        t = int(self.total.item())
        return self.idx1[:t].cpu().numpy(), self.pos2[:t].cpu().numpy(), self.neg2[:t].cpu().numpy()


def select_patch_pairs(center_m, center_n, table, npoints, hard_only=False, seed=0, max_records=500, py_random=random):
    """center_m / center_n: (m0,) int32 device tensors (FPS centre indices of the query / positive cloud in original point indices);
    table: OverlapTable (or a sequence of records).  Returns a PairSelection, or None when the table is empty."""
    if not isinstance(table, OverlapTable):
        table = OverlapTable.from_records(table)
    check_device(center_m, center_n)
    k_list = list(range(len(table)))
    if len(k_list) > max_records:
        k_list = py_random.sample(k_list, max_records)               # train_place_recognition.py:330-331 (the host's `random` stream)
    if not k_list:
        return None
    dev = center_m.device
    m0 = center_m.numel()
    idx1, no, nv, fo, fv = table.take(k_list, hard_only)
    cap = int(np.minimum(np.diff(no), m0).sum())
    if cap == 0:
        return None
    packed = torch.from_numpy(np.concatenate([idx1, no, nv, fo, fv])).to(dev, non_blocking=True)       # one H2D copy
    sizes = [len(idx1), len(no), len(nv), len(fo), len(fv)]
    d_idx1, d_no, d_nv, d_fo, d_fv = torch.split(packed, sizes)
    nrec = len(idx1)
    inv = torch.empty(2 * npoints, dtype=torch.int32, device=dev)
    counts = torch.empty(nrec, dtype=torch.int32, device=dev)
    cm, cn = center_m.reshape(-1).int().contiguous(), center_n.reshape(-1).int().contiguous()
    with torch.cuda.device(dev):
        call("pa_patch_pairs_count", nrec, ptr(d_idx1), ptr(d_no), ptr(d_nv), ptr(d_fo), ptr(d_fv), npoints, m0, ptr(cm), ptr(cn), ptr(inv), ptr(counts))
        incl = torch.cumsum(counts, 0, dtype=torch.int32)
        offsets = (incl - counts).contiguous()
        out = torch.zeros(3, cap, dtype=torch.int32, device=dev)
        call("pa_patch_pairs_fill", nrec, ptr(d_idx1), ptr(d_no), ptr(d_nv), ptr(d_fo), ptr(d_fv), npoints, m0, ptr(inv), int(seed) & (2 ** 64 - 1),
             ptr(offsets), ptr(out[0]), ptr(out[1]), ptr(out[2]))
    return PairSelection(out[0], out[1], out[2], incl[-1].long(), cap)


def contrastive_loss_selected(feat_m, feat_n, sel, margin):
    """contrastive_loss (pointnetvlad_loss.py:170-186) over the selected triplets: mean ||q - p||^2 + mean max(margin - ||q - n||, 0)^2,
    means over the `total` valid triplets.  Returns (loss, has_triplets) as device scalars."""
    valid = (torch.arange(sel.cap, device=feat_m.device) < sel.total).to(feat_m.dtype)
    q = feat_m.index_select(0, sel.idx1.long())
    p = feat_n.index_select(0, sel.pos2.long())
    n = feat_n.index_select(0, sel.neg2.long())
    cnt = sel.total.clamp(min=1).to(feat_m.dtype)
    qp = (F.pairwise_distance(q, p).pow(2) * valid).sum() / cnt
    qn = (torch.clamp(margin - F.pairwise_distance(q, n), min=0.0).pow(2) * valid).sum() / cnt
    return qp + qn, (sel.total > 0).to(feat_m.dtype)


def patch_feature_contrast_loss(nn_dict, recon, margin, npoints, hard_only=False, seed=0):
    """The `use_patch_feature_contrast` branch of train_one_epoch (:308-385): nn_dict {(m, n): records}; recon = the model's
    patch_recon_data dict.  Average of the per-pair contrastive losses over the pairs that produced triplets (0 when none did)."""
    where = {c: k for k, c in enumerate(recon["cloud_indices"])}
    total, pairs = 0.0, 0.0
    for j, ((m, n), records) in enumerate(nn_dict.items()):
        if m not in where or n not in where:
            continue
        km, kn = where[m], where[n]
        sel = select_patch_pairs(recon["center_indices"][km].reshape(-1), recon["center_indices"][kn].reshape(-1), records, npoints,
                                 hard_only=hard_only, seed=seed * 1000003 + j)
        if sel is None:
            continue
        loss, has = contrastive_loss_selected(recon["patch_features"][km], recon["patch_features"][kn], sel, margin)
        total = total + loss * has
        pairs = pairs + has
    if isinstance(pairs, float):
        return None
    return total / pairs.clamp(min=1.0)
