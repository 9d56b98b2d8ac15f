"""EMD (16, 4096, 3) at 64 / 1024 auction rounds: persistent one-workgroup-per-cloud kernel vs the chip-wide one-launch-per-round form (python tools/emd_time.py)."""
import sys, os, time, ctypes
sys.path.insert(0, os.getcwd())
import torch
from patchaugnet_amd import emd_module, _lib
lib = _lib.lib()
lib.pa_emd_persistent_enable.argtypes, lib.pa_emd_persistent_enable.restype = [ctypes.c_int], None
g = torch.Generator().manual_seed(11)
p1, p2 = (torch.rand(16, 4096, 3, generator=g).cuda() for _ in range(2))
f = emd_module.emdModule()
for form in (1, 0, -1):
    lib.pa_emd_persistent_enable(form)
    for iters in (64, 1024):
        f(p1, p2, 0.02, iters); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            d, a = f(p1, p2, 0.02, iters)
        torch.cuda.synchronize()
        print({1: "one workgroup per cloud", 0: "chip-wide, launch per round", 2: "chip-wide, resident rounds", -1: "default (chip-wide, launch per round)"}[form], iters, "iters: %.2f ms" % ((time.perf_counter() - t0) / 3 * 1e3), "mean sqrt dist %.6f" % d.sqrt().mean().item())
# the same call replayed from a captured hipGraph (the training step captures its losses): host launch cost out of the picture
lib.pa_emd_persistent_enable(-1)
for iters in (64, 1024):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        f(p1, p2, 0.02, iters)
        gph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gph, stream=s):
            d, a = f(p1, p2, 0.02, iters)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            gph.replay()
        torch.cuda.synchronize()
        print("default form, hipGraph replay", iters, "iters: %.2f ms" % ((time.perf_counter() - t0) / 3 * 1e3), "mean sqrt dist %.6f" % d.sqrt().mean().item())
