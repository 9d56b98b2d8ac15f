"""Out-of-bounds writes of the fused engine's kernels: every torch.empty / torch.zeros the engine issues during one forward gets a sentinel-filled guard
band on both sides; after the forward the bands must be intact.  python tools/probes/engine_guards.py [f16|f32] [patch_aug_net|pptnet]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from patchaugnet_amd import configs, patch_aug_net, pptnet, engine
from patchaugnet_amd.weights import seeded_state_dict, synthetic_submaps
dt = sys.argv[1] if len(sys.argv) > 1 else "f16"
name = sys.argv[2] if len(sys.argv) > 2 else "patch_aug_net"
m = pptnet.Network(param=configs.pptnet_config(), use_normalize=True) if name == "pptnet" else patch_aug_net.Network(param=configs.patch_aug_net_config(), use_a2a_recon=True, use_l2_norm=True)
m.load_state_dict(seeded_state_dict(m.state_dict())); m = m.cuda().eval(); m.mlp_dtype = dt
x = synthetic_submaps(32, 4096, 70).cuda()
with torch.no_grad():
    m(x, return_feat=False); m(x, return_feat=True)
torch.cuda.synchronize()
GUARD = 4096          # bytes on each side
SENT = 0xA5
records = []
real_empty, real_zeros = torch.empty, torch.zeros
import traceback
def guarded(fill):
    def f(*shape, dtype=torch.float32, device=None, **kw):
        if len(shape) == 1 and isinstance(shape[0], (tuple, list, torch.Size)):
            shape = tuple(shape[0])
        if device is None or torch.device(device).type != "cuda":
            return (real_zeros if fill else real_empty)(*shape, dtype=dtype, device=device, **kw)
        n = 1
        for d in shape: n *= int(d)
        nbytes = n * torch.empty((), dtype=dtype).element_size()
        pad = (-nbytes) % 256
        buf = real_empty(GUARD + nbytes + pad + GUARD, dtype=torch.uint8, device=device)
        buf.fill_(SENT)
        mid = buf[GUARD:GUARD + nbytes]
        if fill: mid.zero_()
        t = mid.view(dtype).view(shape)
        where = [fr for fr in traceback.extract_stack() if "engine.py" in fr.filename][-1:]
        records.append((buf, nbytes, pad, f"{tuple(shape)} {dtype} engine.py:{where[0].lineno if where else '?'}"))
        return t
    return f
class TorchProxy:
    def __getattr__(self, k):
        if k == "empty": return guarded(False)
        if k == "zeros": return guarded(True)
        return getattr(torch, k)
engine.torch = TorchProxy()
with torch.no_grad():
    for views in (False, True):
        m(x, return_feat=views)
torch.cuda.synchronize()
engine.torch = torch
bad = 0
for buf, nbytes, pad, desc in records:
    lo = buf[:GUARD]; hi = buf[GUARD + nbytes + pad:]
    blo = int((lo != SENT).sum()); bhi = int((hi != SENT).sum()); bpad = int((buf[GUARD + nbytes:GUARD + nbytes + pad] != SENT).sum())
    if blo or bhi or bpad:
        bad += 1
        idx = (hi != SENT).nonzero().flatten()
        print(f"OVERRUN {desc}: {nbytes} bytes; guard bytes changed: below {blo}, padding {bpad}, above {bhi}" + (f" (first at +{int(idx[0])}, last at +{int(idx[-1])})" if len(idx) else ""))
print(f"{name} {dt}: {len(records)} engine allocations checked, {bad} with damaged guard bands")
