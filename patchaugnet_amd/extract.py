"""Pipelined descriptor extraction over a list of submap batches (the caller-side loop of the hot path).

Counterpart of the batch loop in the reference's ``SceneDataSet.make_descs`` (datasets/scene_dataset.py:510-523,
:666-686: build a (B,1,N,3) tensor, ``model(feed)`` under no_grad, collect the (B,256) descriptors).  The reference
runs batches strictly one after another on one stream.  On MI355X a batch of 32 clouds keeps only 32 of 256 CUs busy
during the ~1200 serial rounds of farthest-point sampling, so consecutive batches are issued round-robin on a few
HIP streams: batch i+1's sampling overlaps batch i's MFMA work.  Results are identical to sequential execution
(every batch is an independent forward pass).
"""
import torch


_PRIMED = set()


def _prime_stream_queues(device):
    """ROCm 7.2 binds a HIP stream to one of the process's 4 hardware queues when the stream is first used.  Measured on MI355X
    (tools/probes/small_batch.py): four pipeline streams that are the FIRST streams of the process sustain 23.0 k submaps/s, the same
    four streams created after a few other streams have run anything sustain 29.5 k (they then spread one per hardware queue instead
    of crowding next to the default stream).  So a handful of throw-away pool streams run one tiny kernel each before the pipeline's
    own streams are taken.  Pure scheduling: no effect on results.  PA_STREAM_PRIME=0 disables it."""
    import os
    key = (device.type, device.index)
    if key in _PRIMED or os.environ.get("PA_STREAM_PRIME", "1") == "0":
        return
    _PRIMED.add(key)
    for _ in range(4):
        with torch.cuda.stream(torch.cuda.Stream(device=device)):
            torch.zeros(8, device=device).add_(1)
    torch.cuda.synchronize(device)


_POOL = {}


def _pipeline_streams(device, n):
    """The process's pipeline streams for `device`: created once (after the priming above) and handed to every StreamPipeline /
    GraphedExtractor.  A second extractor that created four NEW streams landed them on whatever hardware queues the runtime had left: the same
    model measured 57 k submaps/s in a fresh process and 37 k as the sixth extractor of a long-lived one (bench.py's extra configurations,
    round 5).  A captured graph replays on whichever stream is current, so sharing the streams between extractors costs nothing."""
    _prime_stream_queues(device)
    key = (device.type, device.index)
    pool = _POOL.setdefault(key, [])
    while len(pool) < n:
        pool.append(torch.cuda.Stream(device=device))      # equal priorities: one high-priority stream among them cost 12 %
    return pool[:n]


def _prepare(model, device):
    """Build the model's fused engine (folded + packed weights) on the CALLER's current stream before any pipeline stream touches it:
    a lazily built engine would enqueue its pack kernels on whichever pipeline stream runs the first batch, and the other streams would
    read the packed buffers with nothing ordering them after those kernels."""
    prep = getattr(model, "prepare", None)
    if prep is not None and not getattr(model, "training", False) and getattr(model, "fused_eval", True):
        with torch.cuda.device(device):
            prep(device)


class StreamPipeline:
    def __init__(self, n_streams=2, device=None):
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.streams = list(_pipeline_streams(self.device, max(1, n_streams)))
        self._i = 0

    def begin(self):
        """Make every pipeline stream wait for work already queued on the caller's stream (inputs, weights)."""
        cur = torch.cuda.current_stream(self.device)
        for s in self.streams:
            s.wait_stream(cur)

    def submit(self, fn, *args, **kw):
        """Run fn on the next stream of the ring; returns fn's result (tensors are owned by that stream)."""
        s = self.streams[self._i % len(self.streams)]
        self._i += 1
        with torch.cuda.stream(s):
            return fn(*args, **kw)

    def end(self):
        """Make the caller's stream wait for everything submitted."""
        cur = torch.cuda.current_stream(self.device)
        for s in self.streams:
            cur.wait_stream(s)


class GraphedExtractor:
    """One captured hipGraph of ``model(x, return_feat=False)`` per pipeline stream, replayed round-robin.

    A step of the fused engine is ~45 kernel launches; issuing them from Python costs the host 0.30 ms per step, which bounds
    throughput below batch 8 and occupies a CPU per rank.  A graph replay costs 0.03 ms and runs exactly the same kernels on the
    same arguments (bit-identical descriptors, tests/test_gpu_models.py).  Each slot owns static input / output buffers, so
    ``n_streams`` steps can be in flight; a slot's output is valid until that slot runs again (copy it out with ``out=``).
    Only for a fixed batch shape; ragged tails go through the eager path (extract_descriptors does that)."""

    def __init__(self, model, batch_shape, n_streams=4, device=None, warmup=2, resident_inputs=None):
        """resident_inputs: optional list of device tensors of ``batch_shape`` that ALREADY hold the batches (one per slot, reused round-robin;
        a single tensor serves every slot): slot i's graph is captured reading resident_inputs[i % len] IN PLACE, so ``run(that_tensor)``
        replays with no staging copy.  A slot whose bound tensor is its own (len(resident_inputs) >= n_streams, distinct storages) also
        accepts any other tensor: it is copied into the bound buffer first, on the slot's stream (which overwrites the resident tensor).
        Slots that SHARE a bound tensor accept only that tensor: a staging copy into it on one slot's stream would race with the other
        slots' replays reading it on theirs, so run() refuses instead of giving silently wrong descriptors."""
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        assert not model.training, "hipGraph capture is for evaluation (fused engine, no autograd)"
        streams = _pipeline_streams(self.device, max(1, n_streams))
        self.slots = []
        cur = torch.cuda.current_stream(self.device)
        _prepare(model, self.device)                   # BatchNorm folding / weight packing on the caller's stream, before any slot stream runs
        with torch.no_grad():
            for i in range(max(1, n_streams)):
                st = streams[i]
                if resident_inputs:
                    x = resident_inputs[i % len(resident_inputs)]
                    assert x.is_cuda and tuple(x.shape) == tuple(batch_shape) and x.dtype == torch.float32 and x.is_contiguous()
                else:
                    x = torch.zeros(batch_shape, dtype=torch.float32, device=self.device)
                st.wait_stream(cur)
                with torch.cuda.stream(st):
                    for _ in range(warmup):            # weights folded / packed, every lazy buffer allocated before capture
                        model(x, return_feat=False)
                cur.wait_stream(st)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=st, capture_error_mode="thread_local"):   # other threads (RCCL watchdog) may poll events
                    y = model(x, return_feat=False)
                self.slots.append((g, x, y, st))
        torch.cuda.synchronize(self.device)
        ptrs = [sl[1].data_ptr() for sl in self.slots]
        self._shared_input = [ptrs.count(p) > 1 for p in ptrs]       # slot's bound input is read by another slot's graph too
        self._i = 0
        self._model = model
        self._engine = getattr(model, "_engine", None)      # the graphs point at THIS engine's folded / packed weight buffers

    def begin(self):
        eng = getattr(self._model, "_engine", None)
        if eng is not self._engine or self._model.training or (eng is not None and eng.stale(self._model)):
            raise RuntimeError("GraphedExtractor: the model's weights or mode changed after capture (load_state_dict / train()); "
                               "build a new GraphedExtractor")
        cur = torch.cuda.current_stream(self.device)
        for _, _, _, st in self.slots:
            st.wait_stream(cur)

    def run(self, x, out=None):
        """x: (B,1,N,3) device or pinned-host tensor of the captured shape.  Returns the slot's output buffer (or ``out``)."""
        k = self._i % len(self.slots)
        g, xs, ys, st = self.slots[k]
        foreign = x is not xs and not (x.is_cuda and x.data_ptr() == xs.data_ptr())
        if foreign and self._shared_input[k]:
            raise RuntimeError("GraphedExtractor.run: this slot's graph reads a resident input that other slots read too; copying another "
                               "tensor into it would race with their replays.  Pass the resident tensor itself, or build the extractor with "
                               "one distinct resident tensor per stream (or none: every slot then owns a staging buffer)")
        self._i += 1
        with torch.cuda.stream(st):
            if foreign:
                xs.copy_(x, non_blocking=True)
            g.replay()
            if out is not None:
                out.copy_(ys, non_blocking=True)
        return ys if out is None else out

    def end(self):
        cur = torch.cuda.current_stream(self.device)
        for _, _, _, st in self.slots:
            cur.wait_stream(st)


class SampledAheadExtractor:
    """Descriptor extraction over a KNOWN list of batches (the reference's ``SceneDataSet.make_descs`` loop, datasets/scene_dataset.py:510-523, walks a
    dataset whose submaps are all there before the first forward) with the first-level sampling taken out of the per-batch graphs:

      * the farthest-point sampling of a whole GROUP of batches (``group`` x B clouds; the default 16 x 32 = 512 clouds is two sampling workgroups
        per CU, one round of workgroups: half the CU-time per cloud of a 256-cloud launch, csrc/fps.hip launch_reg) is ONE launch per level on a
        sampling stream, a group ahead of the batches that consume it -- ~1000 serial rounds per cloud that depend on coordinates only;
      * the rest of a batch's step is a captured hipGraph per (buffer set, position in the group) on ``n_streams`` feature streams, reading the
        group's coordinates and samples IN PLACE (``PatchAugNetEngine.forward(s0=...)``; graphs of one stream share a memory pool).

    Why: after a synchronisation the four plain graphs of ``GraphedExtractor`` run in lock-step -- four samplings at once
    on 128 of 256 CUs with nothing else to do, then four dense phases sharing the chip -- and need ~16 steps to drift apart; a 20-step region
    (the driver's protocol) runs at 39.3 k submaps/s that way, at 41.5 k with the sampling of 8 batches a group ahead and at 42.1 k with 16; long regions
    41.5 k (plain graphs, groups of 8) against 42.9 k (groups of 16): the sampling launches' CU-time is what the dense phases get back.  Descriptors are bit-identical to the plain forward (tests/test_gpu_extract.py).

    Only for a fixed batch shape and the fused engine; ``GraphedExtractor`` stays the tool for one batch at a time."""

    def __init__(self, model, batch_shape, n_streams=4, group=16, device=None, warmup=1, ahead=None):
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        assert not model.training, "hipGraph capture is for evaluation (fused engine, no autograd)"
        B, _, N, _ = batch_shape
        self.shape, self.group = tuple(batch_shape), int(group)
        streams = _pipeline_streams(self.device, max(1, n_streams))
        self.feat = streams
        # the sampling stream is one stream MORE than the feature streams (it shares a hardware queue with one of them; with one launch per group
        # that costs nothing -- a sampling stream INSTEAD of a feature stream does: 39.2-39.7 k against 41.5 k)
        self.samp = _sampling_stream(self.device)
        cur = torch.cuda.current_stream(self.device)
        _prepare(model, self.device)
        with torch.no_grad():
            x0 = torch.zeros(batch_shape, dtype=torch.float32, device=self.device)
            model(x0, return_feat=False)                                  # builds the engine
            eng = self._engine = model._engine
            # what runs a group ahead: "samplings" (default) = the farthest-point sampling of EVERY level (level i + 1 samples level i's centres: coordinates
            # only); "sampling" = the first level's alone; "geometry" = every launch that depends on coordinates only (+ centre gathers, neighbour search of all
            # levels, the decoder's 3-NN weights).  Measured (profiles/r06_ab_log.txt), 20 steps: 41.3-41.6 k (samplings) / 41.1-41.4 k (sampling) / 39.0-39.8 k
            # (geometry) / 39.2-40.1 k (plain graphs); PPT-Net fp16 62.3 / 61.6 k.  The group's chip-filling SEARCH launches on the side stream take from the dense
            # kernels what they save the graphs; the small sampling launches of the coarser levels do not.
            import os
            self.mode = ahead or os.environ.get("PA_AHEAD", "samplings")
            self.sets = []
            pools = [None] * len(streams)
            for q in range(2):
                xbig = torch.zeros((group,) + self.shape, dtype=torch.float32, device=self.device)
                geo = eng.geometry_buffers(group * B, N, self.device)
                eng.compute_geometry(xbig.view(group * B, N, 3), geo)
                graphs = []
                for p in range(group):
                    k = p % len(streams)
                    st = streams[k]
                    gs = eng.geometry_slice(geo, p * B, (p + 1) * B)
                    kw = {"geo": gs} if self.mode == "geometry" else {"geo": {"cidx": gs["cidx"], "nxyz": gs["nxyz"]}} if self.mode == "samplings" else {"s0": (gs["cidx"][0], gs["nxyz"][0])}
                    st.wait_stream(cur)
                    with torch.cuda.stream(st):
                        for _ in range(warmup if pools[k] is None else 0):
                            eng.forward(xbig[p], views=False, **kw)
                    cur.wait_stream(st)
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, stream=st, pool=pools[k], capture_error_mode="thread_local"):
                        y, _ = eng.forward(xbig[p], views=False, **kw)
                    if pools[k] is None:
                        pools[k] = g.pool()                               # graphs of one stream never overlap: one pool (their outputs are copied out before the next replay)
                    graphs.append((g, y, st))
                self.sets.append((xbig, geo, graphs))
        torch.cuda.synchronize(self.device)
        self._model = model

    def extract(self, batches, out):
        """batches: (nb, B, 1, N, 3) tensor or a sequence of (B, 1, N, 3) tensors (device, or pinned host: copied on the sampling stream);
        out: (nb, B, 256) device tensor.  Returns ``out``; the caller's stream waits for everything before this returns (like ``GraphedExtractor.end``)."""
        eng = getattr(self._model, "_engine", None)
        if eng is not self._engine or self._model.training or (eng is not None and eng.stale(self._model)):
            raise RuntimeError("SampledAheadExtractor: the model's weights or mode changed after capture (load_state_dict / train()); build a new one")
        nb = len(batches)
        if nb == 0:
            return out
        G, B = self.group, self.shape[0]
        N = self.shape[2]
        cur = torch.cuda.current_stream(self.device)
        samp = self.samp
        samp.wait_stream(cur)
        for s in self.feat:
            s.wait_stream(cur)
        ngroups = (nb + G - 1) // G
        consumed = [None, None]               # per buffer set: the end events of the graphs that last read it
        ev_s = [None] * ngroups
        contiguous = torch.is_tensor(batches)

        def sample(gi):
            q = gi % 2
            xbig, geo, _ = self.sets[q]
            n = min(G, nb - gi * G)
            with torch.cuda.stream(samp):
                for e in consumed[q] or ():
                    samp.wait_event(e)
                if contiguous:
                    xbig[:n].copy_(batches[gi * G:gi * G + n], non_blocking=True)
                else:
                    for j in range(n):
                        xbig[j].copy_(batches[gi * G + j], non_blocking=True)
                self._engine.compute_geometry(xbig.view(G * B, N, 3)[:n * B], self._engine.geometry_slice(geo, 0, n * B), first_level_only=self.mode == "sampling", samplings_only=self.mode == "samplings")
                ev_s[gi] = torch.cuda.Event()
                ev_s[gi].record(samp)
        sample(0)
        for gi in range(ngroups):
            q = gi % 2
            n = min(G, nb - gi * G)
            ends = []
            for j in range(n):
                g, y, st = self.sets[q][2][j]
                with torch.cuda.stream(st):
                    st.wait_event(ev_s[gi])
                    g.replay()
                    out[gi * G + j].copy_(y, non_blocking=True)
                    e = torch.cuda.Event()
                    e.record(st)
                    ends.append(e)
                if j == 0 and gi + 1 < ngroups:
                    sample(gi + 1)            # queued right behind this group's first graph: runs under the group's dense kernels
            consumed[q] = ends
        cur.wait_stream(samp)
        for s in self.feat:
            cur.wait_stream(s)
        return out


_SAMP = {}


def _sampling_stream(device):
    key = (device.type, device.index)
    if key not in _SAMP:
        _pipeline_streams(device, 4)                                     # the feature streams take their hardware queues first
        _SAMP[key] = torch.cuda.Stream(device=device)
    return _SAMP[key]


@torch.no_grad()
def extract_descriptors(model, batches, n_streams=4, out=None, graphs=False):
    """batches: iterable of (B,1,N,3) device tensors -> (sum B, 256) descriptors in input order.  graphs=True runs every batch of the first
    batch's shape through the look-ahead pipeline (SampledAheadExtractor: sampling a group ahead, one captured hipGraph per batch) and the
    remaining (ragged) batches eagerly."""
    batches = list(batches)
    total = sum(b.shape[0] for b in batches)
    dev = batches[0].device
    if out is None:
        out = torch.empty(total, 256, device=dev)
    shape = tuple(batches[0].shape)
    _prepare(model, dev)
    use_graphs = graphs and sum(tuple(b.shape) == shape for b in batches) >= 2 * n_streams
    pipe = StreamPipeline(n_streams, dev)
    pipe.begin()
    off = 0
    common, views = [], []
    for x in batches:
        n = x.shape[0]
        dst = out[off:off + n]
        if use_graphs and tuple(x.shape) == shape:
            common.append(x)                      # the common-shape batches go through the look-ahead pipeline below, in order
            views.append(dst)
        else:
            pipe.submit(lambda x=x, dst=dst: dst.copy_(model(x, return_feat=False)))
        off += n
    pipe.end()
    if common:
        SampledAheadExtractor(model, shape, n_streams, device=dev).extract(common, views)
    return out
