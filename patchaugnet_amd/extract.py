"""Pipelined descriptor extraction over a list of submap batches (the caller-side loop of the hot path).

Counterpart of the batch loop in the reference's ``SceneDataSet.make_descs`` (datasets/scene_dataset.py:510-523,
:666-686: build a (B,1,N,3) tensor, ``model(feed)`` under no_grad, collect the (B,256) descriptors).  The reference
runs batches strictly one after another on one stream.  On MI355X a batch of 32 clouds keeps only 32 of 256 CUs busy
during the ~1200 serial rounds of farthest-point sampling, so consecutive batches are issued round-robin on a few
HIP streams: batch i+1's sampling overlaps batch i's MFMA work.  Results are identical to sequential execution
(every batch is an independent forward pass).
"""
import torch


class StreamPipeline:
    def __init__(self, n_streams=2, device=None):
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.streams = [torch.cuda.Stream(device=self.device) for _ in range(max(1, n_streams))]
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


@torch.no_grad()
def extract_descriptors(model, batches, n_streams=3, out=None):
    """batches: iterable of (B,1,N,3) device tensors -> (sum B, 256) descriptors in input order."""
    batches = list(batches)
    total = sum(b.shape[0] for b in batches)
    dev = batches[0].device
    if out is None:
        out = torch.empty(total, 256, device=dev)
    pipe = StreamPipeline(n_streams, dev)
    pipe.begin()
    off = 0
    for x in batches:
        n = x.shape[0]
        dst = out[off:off + n]
        pipe.submit(lambda x=x, dst=dst: dst.copy_(model(x, return_feat=False)))
        off += n
    pipe.end()
    return out
