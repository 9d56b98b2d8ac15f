#!/usr/bin/env python
"""Where does the host-buffer path lose time?  Steps with (a) nothing, (b) H2D only, (c) D2H only, (d) both, (e) both on a copy stream."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from patchaugnet_amd import configs, patch_aug_net
from patchaugnet_amd.extract import StreamPipeline
from patchaugnet_amd.weights import seeded_state_dict, synthetic_submaps
m = patch_aug_net.Network(param=configs.patch_aug_net_config(), use_a2a_recon=True, use_l2_norm=True)
m.load_state_dict(seeded_state_dict(m.state_dict())); m = m.cuda().eval()
S, K, B = 3, 60, 32
host_x = [synthetic_submaps(B, 4096, seed=i).pin_memory() for i in range(6)]
dev_x = [h.cuda() for h in host_x]
host_d = torch.empty(K, B, 256).pin_memory()
dev_d = torch.empty(K, B, 256, device="cuda")
copy_stream = torch.cuda.Stream()

def run(mode):
    pipe = StreamPipeline(S)
    staged = {}
    def one(i):
        j = i % 6
        if mode in ("h2d", "both"):
            xd = host_x[j].to("cuda", non_blocking=True)
        elif mode == "copystream":
            cur = torch.cuda.current_stream()
            with torch.cuda.stream(copy_stream):
                xd = host_x[j].to("cuda", non_blocking=True)
                ev = torch.cuda.Event(); ev.record(copy_stream)
            cur.wait_event(ev)
            xd.record_stream(cur)
        else:
            xd = dev_x[j]
        d = m(xd, return_feat=False)
        if mode in ("d2h", "both", "copystream"):
            host_d[i].copy_(d, non_blocking=True)
        else:
            dev_d[i].copy_(d)
    with torch.no_grad():
        pipe.begin()
        for i in range(9): pipe.submit(one, i)
        pipe.end(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        pipe.begin()
        for i in range(K): pipe.submit(one, i)
        pipe.end(); torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print(f"{mode:11s} {K*B/dt:8.0f} submaps/s  {dt/K*1e3:.3f} ms/step")

for mode in ("none", "h2d", "d2h", "both", "copystream"):
    run(mode)

print("fresh buffers, repeated")
for rep in range(3):
    host_x = [synthetic_submaps(B, 4096, seed=50 + i).pin_memory() for i in range(6)]
    host_d = torch.empty(K, B, 256).pin_memory()
    run("both")
    run("both")
