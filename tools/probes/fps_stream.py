"""Probe: ONE sampling stream + R feature streams (R + 1 = 4 hardware queues).  The sampling stream runs the first-level sampling of every batch
back to back; batch i's remaining kernels (one captured graph per feature slot) wait for its samples by event and run on stream i % R; the
sampling of batch i + R waits for batch i's graph (it reuses that slot's sample buffers).  Same 20-step protocol as bench.py.
python tools/probes/fps_stream.py [R [steps]]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from patchaugnet_amd import configs, patch_aug_net
from patchaugnet_amd.extract import GraphedExtractor, _pipeline_streams
from patchaugnet_amd.weights import seeded_state_dict, synthetic_submaps
R = int(sys.argv[1]) if len(sys.argv) > 1 else 3
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda", 0)
model = patch_aug_net.Network(param=configs.patch_aug_net_config(), use_a2a_recon=True, use_l2_norm=True)
model.load_state_dict(seeded_state_dict(model.state_dict()))
model = model.cuda().eval()
x = synthetic_submaps(32, 4096, seed=1234).cuda()
xyz = x.squeeze(1).contiguous()
descs = torch.empty(K, 32, 256, device="cuda")
with torch.no_grad():
    ref = model(x, return_feat=False).clone()
    eng = model._engine
    m0 = eng.sampling[0]
    streams = _pipeline_streams(dev, 4)
    fps_st, rest_st = streams[3], streams[:R]
    cur = torch.cuda.current_stream()
    slots = []
    NB = int(os.environ.get("NB", "2"))          # sample-buffer sets (and graphs) per feature slot: the sampling stream runs NB * R batches ahead at most
    for k in range(R * NB):
        st = rest_st[k % R]
        buf = (torch.empty(32, m0, dtype=torch.int32, device="cuda"), torch.empty(32, m0, 3, device="cuda"))
        st.wait_stream(cur)
        with torch.cuda.stream(st):
            eng.sample_first_level(xyz, *buf)
            for _ in range(2):
                eng.forward(x, views=False, s0=buf)
        cur.wait_stream(st)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st, capture_error_mode="thread_local"):
            y, _ = eng.forward(x, views=False, s0=buf)
        slots.append((g, y, buf, st))
    torch.cuda.synchronize()
    gx = GraphedExtractor(model, tuple(x.shape), 4, resident_inputs=[x])

    def region_split():
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fps_st.wait_stream(cur)
        for s in rest_st:
            s.wait_stream(cur)
        done = [None] * len(slots)
        for i in range(K):
            k = i % len(slots)
            g, y, buf, st = slots[k]
            with torch.cuda.stream(fps_st):
                if done[k] is not None:
                    fps_st.wait_event(done[k])            # batch i - R has consumed this slot's sample buffers
                eng.sample_first_level(xyz, *buf)
                ev = torch.cuda.Event()
                ev.record(fps_st)
            with torch.cuda.stream(st):
                st.wait_event(ev)
                g.replay()
                descs[i].copy_(y, non_blocking=True)
                done[k] = torch.cuda.Event()
                done[k].record(st)
        cur.wait_stream(fps_st)
        for s in rest_st:
            cur.wait_stream(s)
        torch.cuda.synchronize()
        return K * 32 / (time.perf_counter() - t0)

    def region_base():
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        gx.begin()
        for i in range(K):
            gx.run(x, out=descs[i])
        gx.end()
        torch.cuda.synchronize()
        return K * 32 / (time.perf_counter() - t0)
    for _ in range(3):
        b = sorted(region_base() for _ in range(5)); s = sorted(region_split() for _ in range(5))
        ok = all(torch.equal(descs[i], ref) for i in range(K))
        print(f"steps {K}: 4 full graphs {b[2]:.0f} ({b[0]:.0f}-{b[-1]:.0f})   1 sampling stream + {R} feature streams {s[2]:.0f} ({s[0]:.0f}-{s[-1]:.0f}) submaps/s  bit-identical {ok}")
