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
