"""CPU restatement of the retrieval + recall bookkeeping (TEST INFRASTRUCTURE -- never imported by the product package).

Follows the reference line by line:
  * per (query trip, reference trip) pair: datasets/scene_dataset.py:1016-1099 (SceneDataSet.get_recall_precision)
  * the loop over trip pairs + KDTree per reference trip: datasets/place_recognition_dataset.py:52-70
  * the average over pairs: place_recognition/evaluate.py:173-237 (run)
Pinned by tests/golden/recall.npz, produced by oracle/gen_recall_golden.py from the reference's own class.
"""
import numpy as np
from sklearn.neighbors import KDTree


def indices_in_dataset(records_size_list):
    """scene_dataset.py:116-124"""
    out, s = [], 0
    for n in records_size_list:
        out.append(list(np.arange(s, s + n)))
        s += n
    return out


def pair_recall_precision(global_descs, database_tree, database_indices, query_indices, positives, same_or_all, top_k=25):
    """scene_dataset.py:1016-1099.  positives: dict query_idx_in_dataset -> list of positive indices (get_tuple(...).positive_indices);
    same_or_all = (query_trip_idx == ref_trip_idx or ref_trip_idx == -1) and not skip_trip_itself  (`add_one_more`, :1041)."""
    num_evaluated = 0
    recall, precision = np.zeros(top_k), np.zeros(top_k)
    one_percent_retrieved = 0
    threshold = max(int(round(len(database_indices) / 100.0)), 1)             # :1026 (round half to even)
    real_top_k = top_k + 1
    if threshold + 1 > real_top_k:
        real_top_k = threshold + 1
    states = []
    for q in query_indices:
        true_positives = positives.get(q, [])
        if not true_positives:
            continue
        num_evaluated += 1
        _, indices = database_tree.query(np.array([global_descs[q]]), k=real_top_k)
        found = [database_indices[j] for j in (indices[0][1:] if same_or_all else indices[0])]     # :1058-1063
        found_positive = False
        for j in range(len(found)):
            if j >= top_k:
                break
            if found[j] == q:
                continue
            if found[j] in true_positives:
                if not found_positive:
                    recall[j] += 1
                    found_positive = True
                precision[j] += 1
        state = 2
        if len(set(found[0:threshold]).intersection(set(true_positives))) > 0:
            one_percent_retrieved += 1
            state = 1
        if found[0] in true_positives:
            state = 0
        states.append(state)
    one_percent_recall = 0.0
    if num_evaluated > 0:
        one_percent_recall = (one_percent_retrieved / float(num_evaluated)) * 100
        recall = (np.cumsum(recall) / float(num_evaluated)) * 100
        precision = (np.cumsum(precision) / float(num_evaluated)) * 100 / np.arange(1, top_k + 1, 1)
    return recall, precision, one_percent_recall, num_evaluated - one_percent_retrieved, threshold, states, num_evaluated, len(database_indices)


def get_recall_precision(global_descs, records_size_list, tuples, top_k=25, skip_trip_itself=False, query_trips=None):
    """place_recognition_dataset.py:52-70.  tuples[(q_trip, r_trip)][query_idx] -> list of positive indices."""
    global_descs = np.asarray(global_descs)
    sample_indices = indices_in_dataset(records_size_list)
    out = {}
    ntrips = len(records_size_list)
    for r in range(ntrips):
        database_indices = sample_indices[r]
        tree = KDTree(global_descs[database_indices])
        for q in range(ntrips):
            if skip_trip_itself and q == r:
                continue
            if query_trips is not None and q not in query_trips:
                continue
            pos = {} if (q == r and skip_trip_itself) else tuples.get((q, r), {})
            same = (q == r) and not skip_trip_itself
            out[q, r] = pair_recall_precision(global_descs, tree, database_indices, sample_indices[q], pos, same, top_k)
    return out


def average(recall_dict, top_k=25):
    """evaluate.py:173-237 for a public (not self-collected) dataset: pairs with q == r or no evaluated query are skipped."""
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
