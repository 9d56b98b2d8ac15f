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


@torch.no_grad()
def extract_descriptors(model, batches, n_streams=4, out=None, graphs=False):
    """batches: iterable of (B,1,N,3) device tensors -> (sum B, 256) descriptors in input order.  graphs=True replays a captured
    hipGraph for every batch of the most common shape (GraphedExtractor) and runs the remaining (ragged) batches eagerly."""
    batches = list(batches)
    total = sum(b.shape[0] for b in batches)
    dev = batches[0].device
    if out is None:
        out = torch.empty(total, 256, device=dev)
    shape = tuple(batches[0].shape)
    _prepare(model, dev)
    gx = GraphedExtractor(model, shape, n_streams, dev) if graphs and sum(tuple(b.shape) == shape for b in batches) >= 2 * n_streams else None
    pipe = StreamPipeline(n_streams, dev)
    pipe.begin()
    if gx is not None:
        gx.begin()
    off = 0
    for x in batches:
        n = x.shape[0]
        dst = out[off:off + n]
        if gx is not None and tuple(x.shape) == shape:
            gx.run(x, out=dst)
        else:
            pipe.submit(lambda x=x, dst=dst: dst.copy_(model(x, return_feat=False)))
        off += n
    pipe.end()
    if gx is not None:
        gx.end()
    return out
