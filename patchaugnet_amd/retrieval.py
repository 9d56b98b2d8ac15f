"""Descriptor retrieval + Recall@N bookkeeping (the caller-side step after descriptor extraction).

Counterpart of the reference's evaluation loop:
  * ``PlaceRecognitionDataSet.get_recall_precision``  datasets/place_recognition_dataset.py:52-70   (KDTree per reference trip,
    every query trip against it)
  * ``SceneDataSet.get_recall_precision``             datasets/scene_dataset.py:1016-1099           (per-pair recall / precision /
    top-1% recall, self-match skipping, 1 % threshold with Python's round-half-to-even)
  * the averaging in ``evaluate.run``                 place_recognition/evaluate.py:173-237

MI355X plan: the reference walks the queries one at a time through a CPU KD-tree in 256-D (where a KD-tree degenerates to
brute force anyway).  Here all queries of a trip pair go through ONE brute-force kNN launch on the GPU
(``pa_knn_generic``, exact KNN_CUDA arithmetic: direct sum of squared differences, order (distance, index)); with
``torch.distributed`` initialised the query trips are dealt round-robin to the ranks and the small index blocks are gathered.
Only the integer bookkeeping (which of the k hits is a true positive) stays on the host.
"""
import itertools

import numpy as np
import torch

from . import knn_cuda
from ._lib import call, ptr


def hip_knn(database, queries, k):
    """database (nd, dim), queries (nq, dim) fp32 HIP tensors -> idx (nq, k) int64, 0-based, ascending (distance, index)."""
    _, ind = knn_cuda.knn(database.t().contiguous(), queries.t().contiguous(), k)       # KNN_CUDA layout: (dim, n) -> (k, nq)
    return ind.t().contiguous()


def indices_in_dataset(records_size_list):
    """scene_dataset.py:116-124 -- dataset indices of each trip."""
    out, s = [], 0
    for n in records_size_list:
        out.append(np.arange(s, s + n))
        s += n
    return out


def one_percent_threshold(num_database):
    """scene_dataset.py:1026 -- ``max(int(round(n / 100.0)), 1)``; Python 3 ``round`` is half-to-even."""
    return max(int(round(num_database / 100.0)), 1)


def real_top_k(num_database, top_k):
    """scene_dataset.py:1027-1029 -- neighbours actually queried: top_k + 1, or threshold + 1 if that is larger."""
    return max(top_k + 1, one_percent_threshold(num_database) + 1)


def pair_recall_precision(found, query_indices, positives, num_database, top_k=25):
    """Bookkeeping of scene_dataset.py:1047-1099 for one (query trip, reference trip) pair.

    found[i]: dataset indices retrieved for query_indices[i], nearest first, self-match column already dropped when the
    reference would drop it (:1058-1063).  positives: dict query index -> list of true positives.  Queries without
    positives are not evaluated (:1050-1051).  Returns the reference's 8-tuple; its ``query_results`` entry (per-query
    record dumps for logging) is reduced to the per-query state 0 (top-1 hit) / 1 (top-1 % hit) / 2 (miss)."""
    threshold = one_percent_threshold(num_database)
    recall, precision = np.zeros(top_k), np.zeros(top_k)
    num_evaluated = one_percent_retrieved = 0
    states = []
    for row, q in zip(found, query_indices):
        tp = positives.get(int(q), [])
        if not tp:
            continue
        num_evaluated += 1
        tp_set = set(tp)
        found_positive = False
        for j, idx in enumerate(row[:top_k]):
            if idx == q:
                continue
            if idx in tp_set:
                if not found_positive:
                    recall[j] += 1
                    found_positive = True
                precision[j] += 1
        state = 2
        if tp_set.intersection(int(v) for v in row[:threshold]):
            one_percent_retrieved += 1
            state = 1
        if int(row[0]) in tp_set:
            state = 0
        states.append(state)
    one_percent_recall = 0.0
    if num_evaluated > 0:
        one_percent_recall = (one_percent_retrieved / float(num_evaluated)) * 100
        recall = (np.cumsum(recall) / float(num_evaluated)) * 100
        precision = (np.cumsum(precision) / float(num_evaluated)) * 100 / np.arange(1, top_k + 1, 1)
    return recall, precision, one_percent_recall, num_evaluated - one_percent_retrieved, threshold, states, num_evaluated, num_database


def pair_recall_precision_fast(found, query_indices, positives, num_database, top_k=25, total=None):
    """pair_recall_precision with the per-query loops as array operations (same 8-tuple, same integers).

    A (query row, retrieved index) pair is a true positive iff its code ``row * total + index`` is among the codes of the positive
    lists; everything else is cumulative sums over boolean matrices.  ``total`` = number of records in the dataset (any bound above
    the largest index)."""
    threshold = one_percent_threshold(num_database)
    found = np.asarray(found, dtype=np.int64)
    query_indices = np.asarray(query_indices, dtype=np.int64)
    keep = np.array([bool(positives.get(int(q))) for q in query_indices], dtype=bool)     # queries without positives are not evaluated
    found, query_indices = found[keep], query_indices[keep]
    ne = int(len(query_indices))
    recall, precision = np.zeros(top_k), np.zeros(top_k)
    if ne == 0:
        return recall, precision, 0.0, 0, threshold, [], 0, num_database
    if total is None:
        total = int(max(found.max(initial=0), query_indices.max(initial=0), max(max(v) for v in positives.values() if v))) + 1
    lists = [positives[int(q)] for q in query_indices]
    lens = np.fromiter((len(l) for l in lists), dtype=np.int64, count=ne)
    rows = np.repeat(np.arange(ne, dtype=np.int64), lens)
    cols = np.fromiter(itertools.chain.from_iterable(lists), dtype=np.int64, count=int(lens.sum()))
    pos_codes = np.unique(rows * total + cols)
    codes = np.arange(ne, dtype=np.int64)[:, None] * total + found
    is_tp = np.isin(codes, pos_codes)                                                    # (ne, k)
    counted = is_tp[:, :top_k] & (found[:, :top_k] != query_indices[:, None])            # :1064-1065 the query itself is skipped
    kk = counted.shape[1]
    precision[:kk] = counted.sum(0)
    any_hit = counted.any(1)
    first = counted.argmax(1)[any_hit]
    np.add.at(recall, first, 1)
    in_threshold = is_tp[:, :threshold].any(1)
    one_percent_retrieved = int(in_threshold.sum())
    states = np.where(is_tp[:, 0], 0, np.where(in_threshold, 1, 2)).tolist()
    one_percent_recall = (one_percent_retrieved / float(ne)) * 100
    recall = (np.cumsum(recall) / float(ne)) * 100
    precision = (np.cumsum(precision) / float(ne)) * 100 / np.arange(1, top_k + 1, 1)
    return recall, precision, one_percent_recall, ne - one_percent_retrieved, threshold, states, ne, num_database


class PositiveTable:
    """The true positives of an evaluation set as arrays, built ONCE per dataset (the reference loads ``QueryPosNegTuple.positive_indices``
    from its pickles once, datasets/scene_dataset.py:243-262, and then walks the Python lists per query in every evaluation, :1047-1099):

      * ``pos[i, j]``      bool, dataset index j is a positive of query i (for the reference trip j belongs to);
      * ``has[i, r]``      bool, query i has at least one positive in reference trip r (queries without are not evaluated, :1050-1051).

    get_recall_precision(..., tuples=PositiveTable) then does the bookkeeping of ALL query trips against a reference trip as a handful of array
    operations instead of one Python pass per (query trip, reference trip) pair -- 506 pairs / 65 561 queries of the Oxford-sized set: 72 -> ~10 ms.
    Dense (N x N bits as bytes): built for N <= 12 000 records (144 MB); larger sets keep the per-pair path."""
    MAX_RECORDS = 12000

    def __init__(self, tuples, records_size_list):
        sizes = [int(v) for v in records_size_list]
        n, ntrips = int(sum(sizes)), len(sizes)
        if n > self.MAX_RECORDS:
            raise ValueError(f"PositiveTable: {n} records > {self.MAX_RECORDS} (dense table); pass the tuples dict instead")
        self.sizes, self.n = sizes, n
        self.pos = np.zeros((n, n), dtype=bool)
        self.has = np.zeros((n, ntrips), dtype=bool)
        starts = np.concatenate([[0], np.cumsum(sizes)])
        for (q, r), table in tuples.items():
            keys = [k for k, v in table.items() if v]
            if not keys:
                continue
            lists = [table[k] for k in keys]
            lens = np.fromiter((len(l) for l in lists), dtype=np.int64, count=len(lists))
            rows = np.repeat(np.asarray(keys, dtype=np.int64), lens)
            cols = np.fromiter(itertools.chain.from_iterable(lists), dtype=np.int64, count=int(lens.sum()))
            inside = (cols >= starts[r]) & (cols < starts[r + 1])          # a positive outside reference trip r can never be retrieved from it
            self.pos[rows[inside], cols[inside]] = True
            self.has[np.asarray(keys, dtype=np.int64), r] = True


def _recall_for_reference_trip(found_all, all_q, pair_of_row, npairs, self_rows, table, num_database, top_k, total):
    """Bookkeeping of scene_dataset.py:1047-1099 for EVERY query trip against one reference trip at once.  found_all (nq, k): retrieved dataset
    indices, nearest first; all_q (nq,): the queries' dataset indices, grouped by pair (pair_of_row ascending); self_rows (nq,) bool: rows of the
    pair q == r when the self-match column has to be dropped (:1058-1060).  Returns one 8-tuple per pair, equal to pair_recall_precision's."""
    threshold = one_percent_threshold(num_database)
    nq, k = found_all.shape
    F = found_all
    if self_rows.any():                                  # `add_one_more`: the first hit is the query itself -> columns shift left by one
        F = F.copy()
        F[self_rows, :-1] = F[self_rows, 1:]
        F[self_rows, -1] = -1
    valid = F >= 0
    is_tp = table.pos[all_q[:, None], np.where(valid, F, 0)] & valid                     # (nq, k)
    counted = is_tp[:, :top_k] & (F[:, :top_k] != all_q[:, None])                        # :1064-1065 the query itself is skipped
    kk = counted.shape[1]
    bounds = np.searchsorted(pair_of_row, np.arange(npairs + 1))
    ne = np.diff(bounds)
    # per-pair sums over contiguous row ranges: differences of a running sum (empty pairs give zero rows)
    run = np.zeros((nq + 1, kk), dtype=np.int64)
    np.cumsum(counted, axis=0, out=run[1:])
    prec_counts = np.zeros((npairs, top_k))
    prec_counts[:, :kk] = run[bounds[1:]] - run[bounds[:-1]]
    any_hit = counted.any(1)
    rec_counts = np.bincount(pair_of_row[any_hit] * top_k + counted.argmax(1)[any_hit], minlength=npairs * top_k).reshape(npairs, top_k).astype(np.float64)
    in_thr = is_tp[:, :threshold].any(1)
    opr_counts = np.bincount(pair_of_row[in_thr], minlength=npairs)
    states = np.where(is_tp[:, 0], 0, np.where(in_thr, 1, 2))
    den = np.maximum(ne, 1).astype(np.float64)[:, None]
    recall_all = (np.cumsum(rec_counts, axis=1) / den) * 100
    precision_all = (np.cumsum(prec_counts, axis=1) / den) * 100 / np.arange(1, top_k + 1, 1)
    out = []
    for p in range(npairs):
        n_e = int(ne[p])
        if n_e == 0:
            out.append((np.zeros(top_k), np.zeros(top_k), 0.0, 0, threshold, [], 0, num_database))
            continue
        retrieved = int(opr_counts[p])
        out.append((recall_all[p], precision_all[p], (retrieved / float(n_e)) * 100, n_e - retrieved, threshold, states[bounds[p]:bounds[p + 1]].tolist(), n_e, num_database))
    return out


def _dist_info():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist, dist.get_rank(), dist.get_world_size()
    return None, 0, 1


def get_recall_precision(global_descs, records_size_list, tuples, top_k=25, skip_trip_itself=False, query_trips=None, knn=hip_knn):
    """place_recognition_dataset.py:52-70 over all trip pairs.

    global_descs: (N, D) fp32 tensor on the device the kNN runs on, rows ordered trip by trip; tuples[(q_trip, r_trip)]
    [query index] -> list of positive dataset indices (``QueryPosNegTuple.positive_indices``), or a ``PositiveTable`` built from that
    dict once per dataset (same results; the bookkeeping then runs per reference trip instead of per pair).  Returns
    {(q_trip, r_trip): 8-tuple of scene_dataset.py:1098-1099}.  With torch.distributed initialised every rank searches the
    pairs (q_trip * n_trips + r_trip) % world == rank and the results are exchanged with all_gather_object (kilobytes)."""
    dist, rank, world = _dist_info()
    sample_indices = indices_in_dataset(records_size_list)
    ntrips = len(records_size_list)
    total = int(sum(records_size_list))
    table = tuples if isinstance(tuples, PositiveTable) else None
    trip_of_record = np.repeat(np.arange(ntrips), [int(v) for v in records_size_list])
    mine = {}
    for r in range(ntrips):
        db_idx = sample_indices[r]
        qs = [q for q in range(ntrips)
              if not (skip_trip_itself and q == r) and (query_trips is None or q in query_trips) and (q * ntrips + r) % world == rank]
        if not qs:
            continue
        k = real_top_k(len(db_idx), top_k)
        # every evaluated query of every owned query trip against this reference trip: ONE kNN launch and one device -> host copy
        evaluated = {}
        if table is not None:      # queries of the owned trips with a positive in trip r, ascending = grouped by query trip
            owned = np.zeros(ntrips, dtype=bool)
            owned[qs] = True
            all_q = np.flatnonzero(table.has[:, r] & owned[trip_of_record]).astype(np.int64)
        else:
            for q in qs:
                pos = tuples.get((q, r), {})
                evaluated[q] = np.array([i for i in sample_indices[q] if pos.get(int(i))], dtype=np.int64)
            all_q = np.concatenate([evaluated[q] for q in qs]) if qs else np.zeros(0, dtype=np.int64)
        found_all = np.zeros((0, k), dtype=np.int64)
        if len(all_q):
            database = global_descs[int(db_idx[0]):int(db_idx[-1]) + 1]
            sel = torch.as_tensor(all_q, device=global_descs.device)
            found_all = db_idx[knn(database, global_descs.index_select(0, sel), min(k, len(db_idx))).cpu().numpy()]
        if table is not None:      # all pairs of this reference trip in one pass of array operations
            pair_index = np.full(ntrips, -1, dtype=np.int64)
            pair_index[qs] = np.arange(len(qs))
            trips_of_q = trip_of_record[all_q]
            pair_of_row = pair_index[trips_of_q]
            self_rows = (trips_of_q == r) & (not skip_trip_itself)
            if found_all.shape[1] < k:                                                       # a reference trip shorter than the requested neighbours
                found_all = np.concatenate([found_all, np.full((found_all.shape[0], k - found_all.shape[1]), -1, dtype=np.int64)], 1)
            res_r = _recall_for_reference_trip(found_all, all_q, pair_of_row, len(qs), self_rows, table, len(db_idx), top_k, total)
            for q, res in zip(qs, res_r):
                mine[q, r] = res
            continue
        off = 0
        for q in qs:
            ne = len(evaluated[q])
            found = found_all[off:off + ne]
            off += ne
            if ne == 0:
                found = np.zeros((0, k), dtype=np.int64)
            elif q == r and not skip_trip_itself:          # `add_one_more`: the first hit is the query itself (:1058-1060)
                found = found[:, 1:]
            mine[q, r] = pair_recall_precision_fast(found, evaluated[q], tuples.get((q, r), {}), len(db_idx), top_k, total=total)
    if dist is None:
        return mine
    parts = [None] * world
    dist.all_gather_object(parts, mine)
    out = {}
    for p in parts:
        out.update(p)
    return dict(sorted(out.items(), key=lambda kv: (kv[0][1], kv[0][0])))


def average(recall_dict, top_k=25):
    """evaluate.py:173-237 for a public dataset: skips q == r pairs and pairs without evaluated queries; returns
    (ave_recall[top_k], ave_precision[top_k], ave_one_percent_recall, lost_mean, lost_sum)."""
    recall, precision, count = np.zeros(top_k), np.zeros(top_k), 0
    opr, lost = [], []
    for (q, r), res in recall_dict.items():
        if q == r or res[6] == 0:
            continue
        recall += np.array(res[0])
        precision += np.array(res[1])
        count += 1
        opr.append(res[2])
        lost.append(res[3])
    return recall / count, precision / count, float(np.mean(opr)), float(np.mean(lost)), int(np.sum(lost))


def get_hard_negatives(query_latent_vector, ref_latent_vectors, negative_indices, num_hard_neg=10, knn=hip_knn):
    """``SceneDataSet.__get_hard_negatives`` (datasets/scene_dataset.py:1101-1113): the num_hard_neg negatives nearest to the query
    in descriptor space, nearest first; [] when there are fewer negatives than requested.  The reference builds a KD-tree over
    the (up to 3000 sampled) negatives for every query; here it is one brute-force kNN launch.

    query_latent_vector (D,), ref_latent_vectors (N, D) device tensors; negative_indices: list of row indices."""
    if len(negative_indices) < num_hard_neg:
        return []
    neg = torch.as_tensor(negative_indices, device=ref_latent_vectors.device, dtype=torch.long)
    found = knn(ref_latent_vectors.index_select(0, neg).contiguous(), query_latent_vector.reshape(1, -1).contiguous(), num_hard_neg)
    return neg[found[0]].tolist()


def get_hard_negatives_batch(query_vectors, ref_latent_vectors, negative_index_lists, num_hard_neg=10, knn=hip_knn, chunk=128):
    """The mining refresh of training (train_place_recognition.py:403-406 -> scene_dataset.py:473-492) for a list of queries.

    Candidate sets differ per query.  With the default kNN the whole refresh is ONE launch on the device: the index lists go up once as
    one padded (nq, L) tensor and a wavefront per query selects the num_hard_neg nearest of its own list in (distance, list position)
    order (pa_knn_candidates) -- 1400 queries x 3000 negatives: 1.66 s as one launch per query.  ``chunk`` is kept for callers that pass
    it; it has no effect on the one-launch path.  num_hard_neg > 64 and indices outside the reference set are handled on the host (above / IndexError).  A custom ``knn`` (tests, CPU stand-ins) keeps the per-query path."""
    nq = len(negative_index_lists)
    if knn is not hip_knn or nq == 0:
        return [get_hard_negatives(q, ref_latent_vectors, negs, num_hard_neg, knn) for q, negs in zip(query_vectors, negative_index_lists)]
    if num_hard_neg > 64:       # the one-launch kernel keeps its top-k in one wavefront (k <= 64): larger requests take the per-query path
        return [get_hard_negatives(q, ref_latent_vectors, negs, num_hard_neg, knn) for q, negs in zip(query_vectors, negative_index_lists)]
    lens = np.fromiter((len(l) for l in negative_index_lists), dtype=np.int64, count=nq)
    L = int(lens.max())
    if L < num_hard_neg:
        return [[] for _ in range(nq)]
    idx = np.full((nq, L), -1, dtype=np.int64)
    for i, l in enumerate(negative_index_lists):
        idx[i, :len(l)] = l
    # the kernel trusts its row indices (like the reference's gather): a bad index raises here, as the torch gather of earlier rounds did
    nref = int(ref_latent_vectors.shape[0])
    if idx.max() >= nref or (idx < -1).any():
        raise IndexError(f"get_hard_negatives_batch: negative index outside [0, {nref}) in negative_index_lists")
    dev = ref_latent_vectors.device
    idx_d = torch.from_numpy(idx).to(dev)
    q_all = query_vectors if torch.is_tensor(query_vectors) else torch.stack(list(query_vectors))
    q_all = q_all.reshape(nq, -1).to(dev).float().contiguous()
    ref = ref_latent_vectors.float().contiguous()
    picked = torch.empty((nq, num_hard_neg), dtype=torch.int64, device=dev)
    # one launch: a wavefront per query walks its own candidate list (csrc/knn_generic.hip, pa_knn_candidates) with the direct-sum
    # arithmetic of the per-query kNN, so the batch returns exactly what the per-query launches return
    with torch.cuda.device(dev):
        call("pa_knn_candidates", ptr(ref), ref.shape[1], ptr(q_all), nq, ptr(idx_d), L, num_hard_neg, ptr(picked))
    picked = picked.cpu().numpy()
    return [picked[i].tolist() if lens[i] >= num_hard_neg else [] for i in range(nq)]
