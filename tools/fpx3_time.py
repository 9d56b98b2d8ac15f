"""Launch time of the finest FP level's chain: exact fp32 MFMA kernel vs the opt-in split-fp16 ("f32x3") kernel; B = 32, HIP events, 50 launches, best of 5."""
import sys
import torch
sys.path.insert(0, ".")
from patchaugnet_amd.engine import _Chain
from tests.test_gpu_chain import make_layers

B, n, m, c2, c1 = 32, 4096, 1024, 256, 3
_, eng = make_layers([c2 + c1, 256, 256, 256], seed=3)
g = torch.Generator().manual_seed(1)
known = torch.randn(B, m, c2, generator=g).cuda()
skip = torch.randn(B, n, c1, generator=g).cuda()
base = torch.randint(0, m, (B, n // 4, 1), generator=g).repeat_interleave(4, 1)
idx3 = ((base + torch.randint(0, 8, (B, n, 3), generator=g)) % m).int().cuda()
w3 = torch.rand(B, n, 3, generator=g)
w3 = (w3 / w3.sum(-1, keepdim=True)).cuda().contiguous()
gk = torch.randn(B * m, 256, device="cuda")
for name, x3 in (("fp32 MFMA (chain_kernel<1,16,FPX>)", False), ("split fp16 operands (fpx3_kernel)", True)):
    ch = _Chain(eng)
    ch.build_premul(c2, c1, x3=x3)
    best = 1e9
    for rep in range(5):
        ch.fp_premul(known, idx3, w3, skip, B, n, m, c2, c1, g_pre=gk)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            ch.fp_premul(known, idx3, w3, skip, B, n, m, c2, c1, g_pre=gk)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 50 * 1e3)
    print(f"{name}: {best:.1f} us per launch", flush=True)
# phase stamps of the split kernel (profiling build: same code + seven LDS stamp stores per wave)
import ctypes
import numpy as np
from patchaugnet_amd import _lib
lib = _lib.lib()
lib.pa_chain_debug_buffer.argtypes = [ctypes.c_void_p]
lib.pa_chain_debug_buffer.restype = None
buf = torch.zeros(512 * 8, dtype=torch.int64, device="cuda")
lib.pa_chain_debug_buffer(ctypes.c_void_p(buf.data_ptr()))
ch.fp_premul(known, idx3, w3, skip, B, n, m, c2, c1, g_pre=gk)
torch.cuda.synchronize()
lib.pa_chain_debug_buffer(None)
t = buf.view(512, 8).cpu().numpy()[:, :7]
t = t[t[:, 6] != 0]
d = (t[:, 1:] - t[:, :-1]) & 0xffffffff
names = ["setup+idx", "gather", "wait W", "layer2", "layer3", "store"]
print(f"fpx3 ({len(t)} wave tiles, cycles): " + "  ".join(f"{nm} {np.median(d[:, i]):.0f}" for i, nm in enumerate(names)), " total", np.median((t[:, 6] - t[:, 0]) & 0xffffffff), flush=True)
