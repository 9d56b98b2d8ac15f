#!/usr/bin/env python
"""Untraced picture of the extraction pipeline: one (start, end) HIP event pair around every graph replay on its stream + the host time of every
submission.  python tools/graph_timeline.py [streams [steps]]   -> per step: host submit time, GPU start / end (us from the first start), stream"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from patchaugnet_amd import configs, patch_aug_net
from patchaugnet_amd.extract import GraphedExtractor
from patchaugnet_amd.weights import seeded_state_dict, synthetic_submaps
S = int(sys.argv[1]) if len(sys.argv) > 1 else 4
K = int(sys.argv[2]) if len(sys.argv) > 2 else 40
model = patch_aug_net.Network(param=configs.patch_aug_net_config(), use_a2a_recon=True, use_l2_norm=True)
model.load_state_dict(seeded_state_dict(model.state_dict()))
model = model.cuda().eval()
x = synthetic_submaps(32, 4096, seed=1234).cuda()
descs = torch.empty(K, 32, 256, device="cuda")
with torch.no_grad():
    gx = GraphedExtractor(model, tuple(x.shape), S, resident_inputs=[x])
    for rep in range(3):
        ev, host = [], []
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        gx.begin()
        for i in range(K):
            k = gx._i % len(gx.slots)
            st = gx.slots[k][3]
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            h0 = time.perf_counter()
            a.record(st)
            gx.run(x, out=descs[i])
            b.record(st)
            host.append(((h0 - t0) * 1e6, (time.perf_counter() - t0) * 1e6))
            ev.append((a, b, k))
        gx.end()
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) * 1e6
print(f"streams {S}, {K} steps, wall {wall:.0f} us = {wall / K:.1f} us per step (with the event records)")
base = ev[0][0]
print(" step stream  host_submit_us  host_return_us   gpu_start_us   gpu_end_us   duration_us  idle_before_on_stream_us")
last_end = {}
for i, ((a, b, k), (h0, h1)) in enumerate(zip(ev, host)):
    s, e = base.elapsed_time(a) * 1e3, base.elapsed_time(b) * 1e3
    idle = s - last_end[k] if k in last_end else float("nan")
    last_end[k] = e
    print(f"{i:5d} {k:6d} {h0:15.0f} {h1:15.0f} {s:14.0f} {e:12.0f} {e - s:12.0f} {idle:12.0f}")
