"""Zero-filled device buffers of one training step out of ONE filled allocation (used by train_ops.py and pointops.py's backward kernels)."""
import contextlib
import threading

import torch


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _current_raw_stream(device):
    """hipStream_t of torch's current stream on `device` as an int.  torch.cuda.current_stream() costs ~8 us of Python per call; the raw accessor
    is one C call -- zeros() runs ~80 times per eager training step."""
    if _raw_stream is not None and device.index is not None:
        return _raw_stream(device.index)
    return torch.cuda.current_stream(device).cuda_stream


class _ZeroArena:
    active = False
    device = None
    buf = None          # uint8 storage of this step, zero-filled by ONE launch
    off = 0             # bytes handed out
    used = 0            # bytes asked for in this step (the next step's size)
    demand = {}         # device -> bytes
    # A second, small buffer for tensors that OUTLIVE the step (parameter gradients: zeros(..., keep=True)): a .grad that is a view of the main
    # buffer would pin its activation-sized scatter targets until the gradients are dropped (for good under zero_grad(set_to_none=False) /
    # gradient accumulation, where the first step's tensors are accumulated into forever).
    kbuf = None
    koff = 0
    kused = 0
    kdemand = {}
    owner = None        # raw stream the step was opened on: the fill launch is ordered on THAT stream only
    lock = threading.Lock()


_arena = _ZeroArena()


@contextlib.contextmanager
def zero_arena(device):
    """Within the context (one training step: forward + backward), zeros() hands out slices of ONE buffer that a single launch has filled
    instead of filling ~80 small buffers one launch each (BatchNorm statistics, split-K / atomic accumulators, scatter targets of the
    backward kernels).  The buffer is allocated afresh per step with the size the previous step asked for (the first step fills its
    buffers one by one and only measures); tensors that outlive the step -- parameter gradients, zeros(..., keep=True) -- come out of a second,
    small buffer and keep only THAT alive through their storage, so nothing is ever handed out twice and no .grad pins the activation-sized part.  Requests beyond the buffer (a changed shape) fall back to torch.zeros."""
    a = _arena
    device = torch.device(device)
    if device.type == "cuda" and device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    if a.active or device.type != "cuda":
        yield
        return
    a.active, a.device, a.off, a.used, a.koff, a.kused = True, device, 0, 0, 0, 0
    a.owner = _current_raw_stream(device)
    want, kwant = a.demand.get(device, 0), a.kdemand.get(device, 0)
    a.buf = torch.zeros(want, dtype=torch.uint8, device=device) if want else None
    a.kbuf = torch.zeros(kwant, dtype=torch.uint8, device=device) if kwant else None
    try:
        yield
    finally:
        a.demand[device], a.kdemand[device] = a.used, a.kused
        a.buf, a.kbuf, a.active = None, None, False


def zeros(shape, dtype, device, keep=False):
    """torch.zeros(shape, dtype=dtype, device=device), out of the step's zero-filled buffer when one is open (zero_arena).  keep=True: the tensor
    outlives the step (a parameter gradient) and comes out of the step's small second buffer."""
    a = _arena
    # slices only on the stream the step's fill launch was issued on: another stream (a pipeline worker, an autograd node replayed on a side
    # stream) is not ordered after that launch and gets a buffer of its own.  The THREAD may differ -- autograd runs a device's backward nodes on
    # its own worker thread, on the forward's stream -- so the bookkeeping is under a lock and the ordering is the stream's.
    if a.active and torch.device(device) == a.device and a.owner is not None and a.owner == _current_raw_stream(a.device):
        n = dtype.itemsize
        for d in shape:
            n *= d
        span = (n + 255) // 256 * 256
        with a.lock:
            if keep:
                a.kused += span
                if a.kbuf is not None and a.koff + span <= a.kbuf.numel() and n > 0:
                    t = a.kbuf[a.koff:a.koff + n].view(dtype).view(shape)
                    a.koff += span
                    return t
            else:
                a.used += span
                if a.buf is not None and a.off + span <= a.buf.numel() and n > 0:
                    t = a.buf[a.off:a.off + n].view(dtype).view(shape)
                    a.off += span
                    return t
    return torch.zeros(shape, dtype=dtype, device=device)
