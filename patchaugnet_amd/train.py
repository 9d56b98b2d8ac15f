"""One training step around the descriptor path (BASELINE.json configs[3]).

Mirrors ``run_model`` (place_recognition/train_place_recognition.py:142-164) and the loss assembly of ``train_one_epoch``
(:255-392): the tuple (query, positives, negatives, other negative) is concatenated to one (T, 1, N, 3) batch, pushed through
the model (module path: autograd over the HIP point ops' backward kernels), descriptors are split back into the four
groups for the quadruplet loss, and the patch-reconstruction branch feeds the HIP Chamfer loss.  The contrastive patch-feature term
(:308-385) takes the overlap tables that ride in nn_dict's values (the reference's Uint32Pair records, or any objects / dicts with the
same four fields) and runs on the device (patchaugnet_amd/patch_pairs.py).
"""
import numpy as np
import torch

from . import losses, patch_pairs, train_ops

DEFAULTS = {  # configs/patch_aug_net.yaml:55-75 (training section)
    "TRAIN_POSITIVES_PER_QUERY": 2, "TRAIN_NEGATIVES_PER_QUERY": 14, "MARGIN_1": 0.5, "MARGIN_2": 0.2,
    "TRIPLET_USE_BEST_POSITIVES": False, "LOSS_LAZY": True, "LOSS_IGNORE_ZERO_BATCH": False, "FEATURE_OUTPUT_DIM": 256,
}


def _as_tensor(a):
    return torch.from_numpy(a).float() if isinstance(a, np.ndarray) else a.float()


def run_model(model, queries, positives, negatives, other_neg, nn_dict=None, num_points=4096, require_grad=True, device=None, args=DEFAULTS,
              geometry=None, input_grad=False):
    """train_place_recognition.py:142-164.  queries (bs,1,N,3), positives (bs,P,N,3), negatives (bs,Nn,N,3), other_neg (bs,1,N,3);
    returns {'global_desc': (q, pos, neg, other) split along dim 1, 'patch_recon': dict or None}.

    input_grad: the reference marks the FEED as requiring a gradient (:155 ``feed_tensor.requires_grad_(require_grad)``), which makes autograd carry a
    gradient with respect to the input coordinates through every level -- the first layers' dX contractions, the grouping / interpolation scatters of
    the coordinate channels, the 3-NN weight arithmetic -- into a tensor the training loop drops (nothing reads feed.grad; losses and every PARAMETER
    gradient are the same without it: tests/test_gpu_train_full.py holds them against the reference run).  Default off; True restores the
    reference's graph for callers that want d loss / d coordinates."""
    device = device or next(model.parameters()).device
    q = _as_tensor(queries)
    # the four groups go to the device first and are concatenated there (a host-side cat spins the ATen thread pool, hostcpu.py)
    feed = torch.cat([_as_tensor(t).to(device, non_blocking=True) for t in (q, positives, negatives, other_neg)], 1)
    feed = feed.view((-1, 1, num_points, 3)).requires_grad_(bool(require_grad and input_grad))
    kw = {} if geometry is None else {"geometry": geometry}      # coordinate-only launches done ahead of time (GraphedTrainer(prefetch=True))
    with torch.set_grad_enabled(require_grad):
        out = model(feed, nn_dict, return_feat=False, **kw) if nn_dict is not None else model(feed, return_feat=False, **kw)
    desc, recon = out if nn_dict is not None else (out, None)
    desc = desc.view(q.shape[0], -1, args["FEATURE_OUTPUT_DIM"])
    split = torch.split(desc, [1, args["TRAIN_POSITIVES_PER_QUERY"], args["TRAIN_NEGATIVES_PER_QUERY"], 1], dim=1)
    return {"global_desc": split, "patch_recon": recon}


def training_step(model, optimizer, queries, positives, negatives, other_neg, nn_dict=None, num_points=4096, args=DEFAULTS,
                  loss_alpha=None, place_loss="quadruplet", recon_loss="patch_chamfer", use_patch_feature_contrast=False, epoch=0,
                  use_hard_negative_patch_mining=False, hard_neg_epoch_for_patch_align=10, step_seed=None):
    """train_one_epoch's body for one batch (:255-392): returns the dict of weighted losses.  use_patch_feature_contrast adds the
    contrastive patch-feature term over nn_dict's overlap tables (:308-385; hard-negative patches only once
    epoch > hard_neg_epoch_for_patch_align with use_hard_negative_patch_mining, :345).  step_seed seeds the device-side draw of the far
    (negative) patches; None (default) = a fresh seed per call derived from torch's seed and a call counter, as the reference draws
    np.random.choice from the global generator on every step (:366) -- a fixed seed would repeat the same negatives every step."""
    if step_seed is None:
        step_seed = next_step_seed()
    loss_alpha = loss_alpha or {"place_recognition": 1.0, "patch_recon_a2a": 1.0, "patch_recon_a2b": 1.0}
    model.train()
    optimizer.zero_grad(set_to_none=True)
    with train_ops.zero_arena(next(model.parameters()).device):   # the step's zero-filled accumulators out of one filled buffer
        out = run_model(model, queries, positives, negatives, other_neg, nn_dict, num_points, True, args=args)
        oq, op, on, oo = out["global_desc"]
        cur = {"place_recognition": losses.get_loss_func(place_loss)(oq, op, on, oo, args["MARGIN_1"], args["MARGIN_2"],
                                                                     use_min=args["TRIPLET_USE_BEST_POSITIVES"], lazy=args["LOSS_LAZY"],
                                                                     ignore_zero_loss=args["LOSS_IGNORE_ZERO_BATCH"])}
        recon = out["patch_recon"]
        if recon is not None and getattr(model, "use_a2a_recon", False):
            cur["patch_recon_a2a"] = losses.get_loss_func(recon_loss)(recon["origin_patches"], recon["reconstructed_patches"])
        if recon is not None and use_patch_feature_contrast:
            a2b = patch_pairs.patch_feature_contrast_loss(nn_dict, recon, args["MARGIN_1"], num_points,
                                                          hard_only=epoch > hard_neg_epoch_for_patch_align and use_hard_negative_patch_mining,
                                                          seed=step_seed)
            if a2b is not None:
                cur["patch_recon_a2b"] = a2b
        total = 0.0
        for k in cur:
            cur[k] = cur[k] * loss_alpha.get(k, 1.0)
            total = total + cur[k]
        if float(total.detach()) > 1e-10:                             # :390-392
            total.backward()
            optimizer.step()
    cur["total"] = total
    return {k: float(v.detach()) for k, v in cur.items()}


_step_counter = [0]


def next_step_seed():
    """Seed of the next step's negative-patch draw: torch's initial seed mixed with a per-process call counter (reproducible under
    torch.manual_seed, different on every call)."""
    _step_counter[0] += 1
    return (int(torch.initial_seed()) * 6364136223846793005 + _step_counter[0] * 1442695040888963407) & (2 ** 63 - 1)


def hard_negative_refresh_due(count, batch_size, epoch, hard_neg_epoch, use_hard_neg=True):
    """train_place_recognition.py:401-406 -- after `count` batches of this epoch: re-extract every training submap's descriptor once the
    model is robust enough (epoch > hard_neg_epoch), every 1400 // batch_size batches, at phase 29."""
    return bool(epoch > hard_neg_epoch and use_hard_neg and count % (1400 // batch_size) == 29)


@torch.no_grad()
def update_global_descs(model, load_batch, n_total, batch_size=36, save_dirs=None, n_streams=4, norm_metas=None):
    """``PlaceRecognitionDataSet.update_global_descs`` -> ``SceneDataSet.make_descs`` (place_recognition_dataset.py:37-39,
    scene_dataset.py:494-711) as the training loop uses it (:403-406, batch_size 36): descriptors of all n_total submaps through the fused
    HIP engine (sharded over the ranks when a process group is up: patchaugnet_amd/distributed.py), the model left in the mode it was in.
    save_dirs = (g_desc_dir, l_desc_dir) also writes the reference's per-submap pickle cache (patchaugnet_amd/io.py): ONE pass, every rank
    writes the files of its own shard only (with more than one rank the cache is complete only on a filesystem all ranks share); norm_metas(lo, hi) -> the list of {'scale', 'trans'} dicts of records lo..hi-1 (or a list
    indexed by record; None = un-normalised submaps, identity meta).  Returns the (n_total, 256) matrix on the device: the input of
    retrieval.get_hard_negatives_batch."""
    from .distributed import all_gather_descriptors, dist_info, extract_dataset, shard_bounds
    from .io import save_descriptor_cache
    was_training = model.training
    model.eval()
    try:
        if save_dirs is None:
            return extract_dataset(model, load_batch, n_total, batch_size=batch_size, n_streams=n_streams)
        _, rank, world = dist_info()
        lo, hi = shard_bounds(n_total, rank, world)
        dev = next(model.parameters()).device
        blocks = []
        for b0 in range(lo, hi, batch_size):
            b1 = min(b0 + batch_size, hi)
            x = load_batch(b0, b1)
            if x.device != dev:                      # pinned / host batches, like extract_dataset's loop
                x = x.to(dev, non_blocking=True)
            d, fp, ci = model(x)
            metas = None if norm_metas is None else (norm_metas(b0, b1) if callable(norm_metas) else norm_metas[b0:b1])
            save_descriptor_cache(save_dirs[0], save_dirs[1], b0, d, x, fp, ci, norm_metas=metas)
            blocks.append(d)
        local = torch.cat(blocks, 0) if blocks else torch.empty((0, DEFAULTS["FEATURE_OUTPUT_DIM"]), device=dev)
        return all_gather_descriptors(local, n_total)
    finally:
        model.train(was_training)


class GraphedTrainer:
    """The whole training step -- forward (module path), quadruplet + patch-Chamfer losses, backward, optimizer step -- captured ONCE into
    a hipGraph and replayed per batch (train_one_epoch's body, train_place_recognition.py:255-392, for a fixed tuple shape and a fixed
    set of nn_dict keys).  An eager step issues ~700 launches from Python; on a slow or shared host that, not the GPU, sets the step
    time.  Every launch of the step is capturable: the HIP ops run on the capture stream through the C ABI, nothing synchronises.

    What changes against ``training_step``: the kNN permutation of ``QueryAndGroup_Edge`` (``torch.randperm`` on the host, pointops.py:553)
    is drawn on the host per step and copied into a device buffer the graph reads; the zero-loss branch of :390-392 (skip backward when
    the summed loss is <= 1e-10) is dropped -- with the reconstruction term in the sum the loss is never zero; the optimizer must be
    built with ``capturable=True``; the contrastive patch-feature term (host-side table packing per step) is not part of the graph.
    Losses come back as device scalars (no synchronisation unless the caller reads them)."""

    def __init__(self, model, optimizer, queries, positives, negatives, other_neg, nn_dict, num_points=4096, args=DEFAULTS, loss_alpha=None,
                 place_loss="quadruplet", recon_loss="patch_chamfer", warmup=3, prefetch=False):
        """prefetch=True: sampling, neighbour search and 3-NN weights -- everything that depends on the coordinates but not on the weights,
        ~1 ms of an 8 ms step, most of it the serial furthest point sampling on 18 CUs -- are captured as a SECOND graph and replayed for the
        NEXT batch on a side stream while the current batch trains (``step(..., next_batch=...)``); two buffer sets alternate.  The step
        graph then starts from the prefetched indices.  Same arithmetic, same losses and parameters as prefetch=False.  Each buffer set's graph
        owns its gradient tensors and its optimizer launches read exactly those; ``p.grad`` keeps referencing the tensors of the set captured
        LAST, so code that inspects ``p.grad`` between steps (logging, clipping) sees every second step's gradients only -- such code belongs
        in the captured body or with prefetch=False."""
        from . import pointops
        self.model, self.optimizer, self.args, self.num_points = model, optimizer, dict(args), num_points
        self.nn_dict = nn_dict
        self.loss_alpha = loss_alpha or {"place_recognition": 1.0, "patch_recon_a2a": 1.0}
        self.place_loss, self.recon_loss = place_loss, recon_loss
        dev = next(model.parameters()).device
        assert dev.type == "cuda", "GraphedTrainer runs on the MI355X"
        assert all(g.get("capturable", True) for g in optimizer.param_groups), "build the optimizer with capturable=True (Adam / AdamW ...)"
        self.device = dev
        self.prefetch = bool(prefetch) and hasattr(getattr(model, "backbone", None), "geometry")
        nsets = 2 if self.prefetch else 1
        self.statics = [[_as_tensor(t).to(dev).clone() for t in (queries, positives, negatives, other_neg)] for _ in range(nsets)]
        self.static = self.statics[0]
        self.groupers = [m for m in model.modules() if isinstance(m, pointops.QueryAndGroup_Edge) and m.radius is None and m.knn_dilation > 1]
        self._perm_sets = [[torch.randperm(g.nsample).to(dev) for g in self.groupers] for _ in range(nsets)]
        self._perms = self._perm_sets[0]
        model.train()
        cur = torch.cuda.current_stream(dev)
        side = torch.cuda.Stream(device=dev)
        self._side = side
        self._set_perm_buffers(0)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            for _ in range(warmup):                    # allocator pools, lazy buffers and the optimizer state exist before capture
                optimizer.zero_grad(set_to_none=True)
                self._body(0, self._geometry(0) if self.prefetch else None)
        cur.wait_stream(side)
        torch.cuda.synchronize(dev)
        self.graphs, self.geo_graphs, self.geo, self.losses_k = [], [], [], []
        try:
            for k in range(nsets):
                self._set_perm_buffers(k)
                geo = None
                if self.prefetch:
                    gg = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(gg, capture_error_mode="thread_local"):
                        geo = self._geometry(k)
                    self.geo_graphs.append(gg)
                    self.geo.append(geo)
                optimizer.zero_grad(set_to_none=True)  # the captured backward then CREATES the gradients inside this graph's pool
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    self.losses_k.append(self._body(k, geo))
                self.graphs.append(g)
        finally:
            for gr in self.groupers:                   # the graphs have the buffers' addresses; eager forwards of the model draw their own again
                gr.perm_buffer = None
        # the graphs read the model's related-cloud index tensor(s) by address: hold them for the graphs' lifetime (patch_aug_net.related_index)
        self._pinned = list(getattr(model, "_related_cache", {}).values())
        self.graph, self.losses = self.graphs[0], self.losses_k[0]
        self._k = 0                                    # buffer set of the next step
        self._ready = None                             # (set, ids of the batch tensors) whose geometry the side stream has produced
        self._ev_side = None
        self._captured_hyper = self._hyper()           # the launch constants the graphs were captured with (_check_hyperparameters)
        torch.cuda.synchronize(dev)

    def _set_perm_buffers(self, k):
        for g, buf in zip(self.groupers, self._perm_sets[k]):
            g.perm_buffer = buf                        # read by the forward during warm-up and capture only (reset after capture)

    def _feed(self, k):
        return torch.cat(self.statics[k], 1).view((-1, 1, self.num_points, 3))

    def _geometry(self, k):
        return self.model.backbone.geometry(self._feed(k).squeeze(1))

    def _body(self, k=0, geometry=None):
        q, p, n, o = self.statics[k]
        with train_ops.zero_arena(self.device):
            out = run_model(self.model, q, p, n, o, self.nn_dict, self.num_points, True, device=self.device, args=self.args, geometry=geometry)
            oq, op, on, oo = out["global_desc"]
            a = self.args
            cur = {"place_recognition": losses.get_loss_func(self.place_loss)(oq, op, on, oo, a["MARGIN_1"], a["MARGIN_2"], use_min=a["TRIPLET_USE_BEST_POSITIVES"],
                                                                              lazy=a["LOSS_LAZY"], ignore_zero_loss=a["LOSS_IGNORE_ZERO_BATCH"])}
            recon = out["patch_recon"]
            if recon is not None and getattr(self.model, "use_a2a_recon", False):
                cur["patch_recon_a2a"] = losses.get_loss_func(self.recon_loss)(recon["origin_patches"], recon["reconstructed_patches"])
            total = 0.0
            for k2 in cur:
                cur[k2] = cur[k2] * self.loss_alpha.get(k2, 1.0)
                total = total + cur[k2]
            total.backward()
        self.optimizer.step()
        cur["total"] = total
        return {k2: v.detach() for k2, v in cur.items()}

    def _load(self, k, batch):
        for dst, src in zip(self.statics[k], batch):
            dst.copy_(_as_tensor(src), non_blocking=True)
        # the host permutations are pageable temporaries: kept referenced until this set is loaded again, so that an asynchronous copy that has
        # not been staged yet can never read freed host memory (whatever the runtime's staging policy for small unpinned copies is)
        host = [torch.randperm(g.nsample) for g in self.groupers]
        for hp, buf in zip(host, self._perm_sets[k]):
            buf.copy_(hp, non_blocking=True)
        self.__dict__.setdefault("_perm_host", {})[k] = host

    def step(self, queries, positives, negatives, other_neg, next_batch=None):
        """Copy the batch into the graph's input buffers, draw the step's kNN permutations, replay.  Returns the dict of weighted losses
        (device scalars owned by the graph: valid until that buffer set's next step).  prefetch=True: next_batch = the (queries, positives,
        negatives, other_neg) of the FOLLOWING call; its coordinate-only launches run on a side stream under this step.  A call whose batch
        was not announced that way computes them first, on the main stream.  An announced batch is COPIED at the announcement (inputs and
        geometry stay consistent): the following call must pass the same tensor objects and is trained on their contents as announced.
        The copies are asynchronous on the side stream: pinned host sources must stay untouched until that following call has been
        issued (pageable sources are staged by the runtime before the copy call returns)."""
        batch = (queries, positives, negatives, other_neg)
        self._check_hyperparameters()
        if not self.prefetch:
            self._load(0, batch)
            self.graph.replay()
            return self.losses
        main = torch.cuda.current_stream(self.device)
        k = self._k
        # the announced batch is held by strong reference and matched by identity (an id() of a dropped array can be reused by a new one)
        if self._ready is not None and self._ready[0] == k and all(a is b for a, b in zip(self._ready[1], batch)):
            main.wait_event(self._ev_side)             # the side stream has filled this set's inputs, permutations and geometry
        else:
            if self._ev_side is not None:
                main.wait_event(self._ev_side)         # never overlap a stale prefetch into the set we are about to write
            self._load(k, batch)
            self.geo_graphs[k].replay()
        self._ready = None
        ev = torch.cuda.Event()
        ev.record(main)                                # everything before this step (the other set's previous step) is ordered before it
        self.graphs[k].replay()
        if next_batch is not None:
            o = 1 - k
            self._side.wait_event(ev)
            with torch.cuda.stream(self._side):
                self._load(o, next_batch)
                self.geo_graphs[o].replay()
                self._ev_side = torch.cuda.Event()
                self._ev_side.record(self._side)
            self._ready = (o, tuple(next_batch))
        self._k = 1 - k
        self.losses = self.losses_k[k]
        return self.losses

    def _hyper(self):
        return [{k: v for k, v in g.items() if k != "params" and isinstance(v, (int, float, tuple, bool))} for g in self.optimizer.param_groups]

    def _check_hyperparameters(self):
        """A captured step replays the launch constants it was captured with.  An optimizer that keeps its learning rate on the device
        (patchaugnet_amd.optim.Adam) takes a scheduler's new value here; any other hyper-parameter that changed since the capture -- or a
        changed learning rate of an optimizer without that hook (torch's capturable Adam with a float lr) -- would be silently ignored by the
        replay, so it raises instead."""
        now = self._hyper()
        if now == self._captured_hyper:
            return
        sync = getattr(self.optimizer, "sync_hyperparameters", None)
        for a, b in zip(now, self._captured_hyper):
            changed = {k for k in set(a) | set(b) if a.get(k) != b.get(k)}
            if changed - ({"lr"} if sync is not None else set()):
                raise RuntimeError(f"GraphedTrainer: optimizer hyper-parameters {sorted(changed)} changed after the step was captured; the replay "
                                   "would ignore them -- build a new GraphedTrainer (patchaugnet_amd.optim.Adam follows a changed lr without that)")
        sync()
        self._captured_hyper = now

    def close(self):
        """Nothing to undo on the model (the permutation buffers are the trainer's own); kept for symmetry with GraphedExtractor."""
        self.graph = None
        self.graphs, self.geo_graphs = [], []
