"""Submap loading + normalisation + batch staging: the step in front of the descriptor path.

Reference: ``utils/loading_pointclouds.py:14-38`` (``load_pc_file``: Oxford-style ``.bin`` = raw float64 triples),
``:51-63`` (``normalize_point_cloud``: zero mean, maximum radius 1), and the batch assembly of
``SceneDataSet.make_descs`` (datasets/scene_dataset.py:510-523, :667-670: stack -> float32 -> (B,1,N,3) -> ``.to(device)``).
MI355X side: batches are assembled in pinned host memory and copied with ``non_blocking=True`` on the extraction stream,
so the H2D copy of batch i+1 (1.5 MB per 32 submaps) overlaps batch i's kernels.
"""
import os

import numpy as np
import torch


def load_pc_file(filepath, input_dim=3, num_points=4096, use_np_load=False, dtype=np.float64):
    """loading_pointclouds.py:14-38 (3-D case; the 13-channel hand-crafted-feature variant is not on this path)."""
    if input_dim != 3:
        raise NotImplementedError("only xyz submaps (input_dim=3) are on the descriptor path")
    if use_np_load:
        return np.load(filepath).reshape([-1, 3])
    return np.fromfile(filepath, dtype=dtype).reshape([-1, 3])


def load_pc_files(filenames, dataset_folder, input_dim=3, use_np_load=False, dtype=np.float64):
    """loading_pointclouds.py:41-48 -- silently skips files that do not exist, like the reference."""
    paths = [os.path.join(dataset_folder, f) for f in filenames]
    return [load_pc_file(p, input_dim, use_np_load=use_np_load, dtype=dtype) for p in paths if os.path.exists(p)]


def normalize_point_cloud(pc, return_norm_meta=False, zoom=True):
    """loading_pointclouds.py:51-63 -- subtract the centroid; with zoom divide by the largest point norm."""
    centroid = np.mean(pc, axis=0)
    pc = pc - centroid
    m = 1.0
    if zoom:
        m = np.max(np.sqrt(np.sum(pc ** 2, axis=1)))
        pc = pc / m
    if return_norm_meta:
        return pc, {"scale": m, "trans": centroid}
    return pc


def normalize_point_clouds(pcs, return_norm_meta=False, zoom=True):
    """loading_pointclouds.py:66-78"""
    out, meta = [], []
    for pc in pcs:
        if return_norm_meta:
            p, mm = normalize_point_cloud(pc, True, zoom)
            meta.append(mm)
        else:
            p = normalize_point_cloud(pc, False, zoom)
        out.append(p)
    return (out, meta) if return_norm_meta else out


class BatchStager:
    """(list of (N,3) arrays) -> (B,1,N,3) fp32 device tensor through a ring of pinned host buffers (async H2D)."""

    def __init__(self, batch_size, num_points=4096, device="cuda", depth=3):
        self.device = torch.device(device)
        pin = self.device.type == "cuda"
        self.host = [torch.empty((batch_size, 1, num_points, 3), dtype=torch.float32, pin_memory=pin) for _ in range(depth)]
        self.events = [None] * depth
        self._i = 0

    def stage(self, clouds):
        slot = self._i % len(self.host)
        self._i += 1
        if self.events[slot] is not None:
            self.events[slot].synchronize()                  # the copy that last used this pinned buffer has finished
        h = self.host[slot][:len(clouds)]
        for i, pc in enumerate(clouds):                      # float64 -> float32 conversion happens in this copy (scene_dataset.py:667-669)
            h[i, 0].copy_(torch.from_numpy(np.ascontiguousarray(pc)))
        if self.device.type != "cuda":
            return h.clone()
        d = h.to(self.device, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        self.events[slot] = ev
        return d


# ---- on-disk descriptor cache ------------------------------------------------------------------------------------------------------
# SceneDataSet.make_descs(save=True) writes, per submap index, `<g_desc_dir>/<idx>.pickle` = the (1, C) float32 global descriptor and
# `<l_desc_dir>/<idx>.pickle` = (l_pos (K, 3), l_desc (K, C), norm_meta) -- the K = 1024 FPS centres of the second-finest level and
# their features (datasets/scene_dataset.py:689-707); get_g_desc / get_l_kpt_desc read them back (:784-798, :807-831).  Same file names,
# same pickle protocol, same array shapes and dtypes, so caches are interchangeable with the reference's in both directions.
import pickle


def save_descriptor_cache(g_desc_dir, l_desc_dir, beg_idx, global_descs, feed=None, fp_features=None, center_idx=None, norm_metas=None):
    """scene_dataset.py:689-707.  global_descs (B, C); feed (B, 1, N, 3) or (B, N, 3) = the batch that went into the model;
    fp_features / center_idx = the model's other two outputs (local descriptors = fp_features[-2], centres = center_idx[0]).  The local
    files are written only when center_idx is given, like the reference.  Device tensors are brought to the host in ONE copy each.
    norm_metas: one {'scale', 'trans'} dict per submap (normalize_point_cloud's second result); None = the identity meta the reference
    stores for submaps that were not normalised ({'scale': 1.0, 'trans': zeros (1, 3)}, scene_dataset.py:723), so that
    load_local_descriptor(unify_coord=True) -- and the reference's get_l_kpt_desc -- can always read the file."""
    g = global_descs.detach().cpu().numpy() if torch.is_tensor(global_descs) else np.asarray(global_descs)
    g = np.squeeze(g).reshape([-1, g.shape[-1]])
    os.makedirs(g_desc_dir, exist_ok=True)
    l_pos = l_desc = None
    if center_idx is not None:
        os.makedirs(l_desc_dir, exist_ok=True)
        pts = feed.squeeze(1) if feed.dim() == 4 else feed                                     # (B, N, 3)
        ci = center_idx[0].long()                                                              # (B, K)
        l_pos = torch.gather(pts, 1, ci.unsqueeze(-1).expand(-1, -1, 3)).detach().cpu().numpy()          # index_select per cloud, batched
        l_desc = fp_features[-2].squeeze(-1).permute(0, 2, 1).detach().cpu().numpy()           # (B, K, C)
    for b_i in range(g.shape[0]):
        with open(os.path.join(g_desc_dir, f"{beg_idx + b_i}.pickle"), "wb") as handle:
            pickle.dump(g[b_i].reshape(1, -1), handle, protocol=pickle.HIGHEST_PROTOCOL)
        if l_pos is not None:
            meta = norm_metas[b_i] if norm_metas is not None else {"scale": 1.0, "trans": np.zeros([1, 3])}
            with open(os.path.join(l_desc_dir, f"{beg_idx + b_i}.pickle"), "wb") as handle:
                pickle.dump((l_pos[b_i], l_desc[b_i], meta), handle, protocol=pickle.HIGHEST_PROTOCOL)


def load_global_descriptor(g_desc_dir, idx):
    """scene_dataset.py:784-798 (get_g_desc without the LRU bookkeeping): (1, d) array, or None when the file does not exist."""
    path = os.path.join(g_desc_dir, f"{idx}.pickle")
    if not os.path.exists(path):
        return None
    with open(path, "rb") as handle:
        return pickle.load(handle).reshape(1, -1)


def load_global_descriptors(g_desc_dir, idxs):
    """scene_dataset.py:800-804 (get_g_descs): (len(idxs), d)."""
    return np.concatenate([load_global_descriptor(g_desc_dir, i) for i in idxs], axis=0)


def load_local_descriptor(l_desc_dir, idx, unify_coord=False, global_offset=0.0):
    """scene_dataset.py:807-831 (get_l_kpt_desc without the LRU bookkeeping): (l_kpt (K, 3) float64, l_desc (K, d), norm_meta), or None
    when the file does not exist.  unify_coord maps the key points back to world coordinates: kpt * scale + (trans - global_offset)."""
    path = os.path.join(l_desc_dir, f"{idx}.pickle")
    if not os.path.exists(path):
        return None
    with open(path, "rb") as handle:
        l_kpt, l_desc, norm_meta = pickle.load(handle)
    l_kpt = np.array(l_kpt, dtype=np.float64)
    K = l_kpt.shape[0]
    l_kpt, l_desc = l_kpt.reshape(K, -1), l_desc.reshape(K, -1)
    if unify_coord:
        # the reference reshapes to (1, trans.shape[0]) (scene_dataset.py:827), which only fits the (3,) centroid of normalize_point_cloud and
        # raises on the (1, 3) zeros it stores for un-normalised submaps (:723); reshape(1, -1) reads both
        trans = np.asarray(norm_meta["trans"]).reshape(1, -1) - global_offset
        l_kpt = l_kpt * norm_meta["scale"] + trans
    return l_kpt, l_desc, norm_meta
