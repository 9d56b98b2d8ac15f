#!/usr/bin/env python
"""Phase stamps of the finest FP level's half-K kernel (fpx32_kernel, csrc/fpx_f32.hip FX_STAMP): index loads + first half prologue | layer A (two passes,
the second half prologue between them, bias) | layer B first output half (staging, two passes, store) | second output half.  Cycles per 16-row wave tile.
The shipped library stamps the first 512 tiles (phases and the shader clock: s_memtime ticks per s_memrealtime microsecond; bench.py reports that clock as
roofline.shader_clock_mhz); every tile needs a stamp build:
    tools/build_variant.sh stampall "-DFX_STAMP_TILES=8192" fpx_f32.hip
    FX_STAMP_TILES=8192 PA_LIB_PATH=patchaugnet_amd/csrc/ab/libpa_stampall.so python tools/probes/fx_phases.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from patchaugnet_amd import _lib, configs, patch_aug_net
from patchaugnet_amd.weights import seeded_state_dict, synthetic_submaps

model = patch_aug_net.Network(param=configs.patch_aug_net_config(), use_a2a_recon=True, use_l2_norm=True)
model.load_state_dict(seeded_state_dict(model.state_dict()))
model = model.cuda().eval()
x = synthetic_submaps(32, 4096, seed=1234).cuda()
lib = _lib.lib()
lib.pa_chain_debug_buffer.argtypes = [ctypes.c_void_p]
lib.pa_chain_debug_buffer.restype = None
with torch.no_grad():
    for _ in range(2):
        model(x, return_feat=False)
    chain = model._engine.fp[0]
    NT = int(os.environ.get("FX_STAMP_TILES", "512"))
    buf = torch.zeros(NT * 8, dtype=torch.int64, device="cuda")
    origs = {}
    for nm in ("fp_premul", "fp"):
        def wrapped(*a, _o=getattr(chain, nm), **k):
            lib.pa_chain_debug_buffer(ctypes.c_void_p(buf.data_ptr()))
            r = _o(*a, **k)
            lib.pa_chain_debug_buffer(None)
            return r
        origs[nm] = getattr(chain, nm)
        setattr(chain, nm, wrapped)
    model(x, return_feat=False)
    torch.cuda.synchronize()
    for nm, o in origs.items():
        setattr(chain, nm, o)
t = buf.view(-1, 8).cpu().numpy()
t = t[t[:, 0] > 0]
d = t[:, 1:5] - t[:, 0:4]
print(len(t), "tiles; median cycles per phase:", [int(np.median(d[:, i])) for i in range(4)], "total", int(np.median(t[:, 4] - t[:, 0])))
print("p10 / p90 per phase:", [(int(np.percentile(d[:, i], 10)), int(np.percentile(d[:, i], 90))) for i in range(4)])
print("tile start offsets p0/50/100:", [int(np.percentile(t[:, 0] - t[:, 0].min(), p)) for p in (0, 50, 100)], "end:", [int(np.percentile(t[:, 4] - t[:, 0].min(), p)) for p in (0, 50, 100)])

if (t[:, 5] > 0).all() and (t[:, 6] > t[:, 5]).all():
    mhz = (t[:, 4] - t[:, 0]) / ((t[:, 6] - t[:, 5]) / 100.0)      # shader-clock ticks per microsecond of the 100 MHz real-time counter
    print("shader clock over a tile (s_memtime ticks / s_memrealtime): median %.0f MHz, p10 %.0f, p90 %.0f; tile duration median %.1f us" %
          (np.median(mhz), np.percentile(mhz, 10), np.percentile(mhz, 90), np.median((t[:, 6] - t[:, 5]) / 100.0)))
if len(t) > 4096:
    # the cycle counters of the eight XCDs are not synchronised: tiles [1024 x, 1024 (x + 1)) run on XCD x (the kernel's contiguous tile ranges per XCD)
    full = buf.view(-1, 8).cpu().numpy()
    for x in (0, 3, 7):
        tx = full[1024 * x:1024 * (x + 1)]
        tx = tx[tx[:, 0] > 0]
        t0, t1 = tx[:, 0].min(), tx[:, 4].max()
        edges = np.linspace(t0, t1, 17)
        print(f"XCD {x}: {len(tx)} tiles on 32 CUs = 384 wave slots; launch length {int(t1 - t0)} cycles; sum of tile durations / (384 x length) = {float((tx[:, 4] - tx[:, 0]).sum()) / (384 * (t1 - t0)):.3f}")
        for i in range(16):
            c = 0.5 * (edges[i] + edges[i + 1])
            sel = (tx[:, 0] >= edges[i]) & (tx[:, 0] < edges[i + 1])
            infl = ((tx[:, 0] <= c) & (tx[:, 4] > c)).sum()
            inpro = ((tx[:, 0] <= c) & (tx[:, 1] > c)).sum()
            print(f"  bin {i:2d}: starts {sel.sum():4d}  median duration {int(np.median(tx[sel, 4] - tx[sel, 0])) if sel.any() else 0:7d}  in flight {infl:4d}  in first prologue {inpro:4d}")
