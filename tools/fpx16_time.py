"""Launch time of the finest FP level's fp16 chain (pa_fp_chain_premul_f16) in the three forms of fpx_f16.hip / chain16_kernel.
usage: python tools/fpx16_time.py [B]    (MI355X; prints us per launch: HIP events around 50 back-to-back launches, best of 5)"""
import sys
import torch
sys.path.insert(0, ".")
from patchaugnet_amd import _lib
from patchaugnet_amd.engine import _Chain
from tests.test_gpu_chain import make_layers

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
n, m, c2, c1 = 4096, 1024, 256, 3
_, eng = make_layers([c2 + c1, 256, 256, 256], seed=3)
g = torch.Generator().manual_seed(1)
known = torch.randn(B, m, c2, generator=g).cuda()
skip = torch.randn(B, n, c1, generator=g).cuda()
# neighbours that are spatially coherent like the model's (3-NN of a point among 1024 centres): consecutive rows share neighbours
base = torch.randint(0, m, (B, n // 4, 1), generator=g).repeat_interleave(4, 1)
idx3 = ((base + torch.randint(0, 8, (B, n, 3), generator=g)) % m).int().cuda()
w3 = torch.rand(B, n, 3, generator=g)
w3 = (w3 / w3.sum(-1, keepdim=True)).cuda().contiguous()
import os
gk = torch.randn(B * m, 256, device="cuda")
g16 = gk.half()
os.environ["PA_ENGINE_FPX16"] = "0"
ch = _Chain(eng, f16=True)
ch.build_premul(c2, c1)
pm = ch._premul
import ctypes as _ct
_cast = lambda a: _ct.cast(a, _ct.c_void_p)
from patchaugnet_amd._lib import call, ptr
out = torch.empty(B * n, 256, device="cuda")

def run_g16():
    call("pa_fp_chain_premul_g16", pm["m"], _cast(pm["wpk"]), _cast(pm["bias"]), _cast(pm["kpad"]), _cast(pm["nout"]), B * n, ptr(g16), ptr(idx3),
         ptr(w3), ptr(skip), n, m, 256, c1, ptr(pm["wskip"]), ptr(pm["bias0"]), ptr(out), 256)

def timed(fn):
    best = 1e9
    for rep in range(5):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 50 * 1e3)
    return best

for mode in (8, 4):
    _lib.lib().pa_fpx16_enable(mode)
    print(f"mode {mode}, fp16 table: {timed(run_g16):.1f} us per launch (B = {B})", flush=True)
_lib.lib().pa_fpx16_enable(-1)
g16o = torch.empty(B * m, 256, dtype=torch.float16, device="cuda")
print(f"pa_fp_premul_g16 ({B * m} rows): {timed(lambda: call('pa_fp_premul_g16', B * m, ptr(known), 256, ptr(pm['w1a_p']), ptr(g16o))):.1f} us", flush=True)
gf = torch.empty(B * m, 256, device="cuda")
print(f"pa_linear_f16 pre-multiply: {timed(lambda: call('pa_linear_f16', B * m, 256, 256, ptr(known), 256, ptr(pm['w1a']), ptr(pm['w1a_p']), ptr(pm['zero']), 0, None, 0, ptr(gf), 256)):.1f} us", flush=True)
for mode in (0, 8, 4):
    _lib.lib().pa_fpx16_enable(mode)
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
    print(f"mode {mode}: {best:.1f} us per launch (B = {B})", flush=True)
_lib.lib().pa_fpx16_enable(-1)
# phase stamps of the LDS-shared kernels (profiling build of the kernel: same code + seven LDS stamp stores per wave)
import ctypes
import numpy as np
lib = _lib.lib()
lib.pa_chain_debug_buffer.argtypes = [ctypes.c_void_p]
lib.pa_chain_debug_buffer.restype = None
for mode, tab in ((8, "fp32"), (4, "fp32"), (8, "fp16"), (4, "fp16")):
    lib.pa_fpx16_enable(mode)
    buf = torch.zeros(512 * 8, dtype=torch.int64, device="cuda")
    lib.pa_chain_debug_buffer(ctypes.c_void_p(buf.data_ptr()))
    if tab == "fp16":
        run_g16()
    else:
        ch.fp_premul(known, idx3, w3, skip, B, n, m, c2, c1, g_pre=gk)
    torch.cuda.synchronize()
    lib.pa_chain_debug_buffer(None)
    t = buf.view(512, 8).cpu().numpy()[:, :7]
    t = t[t[:, 6] != 0]
    d = (t[:, 1:] - t[:, :-1]) & 0xffffffff
    names = ["setup+idx", "gather", "wait W", "layer2", "layer3", "store"]
    print(f"mode {mode} {tab} table ({len(t)} wave tiles): " + "  ".join(f"{nm} {np.median(d[:, i]):.0f}" for i, nm in enumerate(names)),
          " total", np.median((t[:, 6] - t[:, 0]) & 0xffffffff), flush=True)
lib.pa_fpx16_enable(-1)
