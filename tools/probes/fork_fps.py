"""Probe: each slot's graph = [first-level sampling of the slot's NEXT batch on a forked branch] || [the rest of the step on the CURRENT batch's
samples]; two graphs per slot ping-pong the sample buffers.  Same 20-step protocol as bench.py.  python tools/probes/fork_fps.py [streams [steps]]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from patchaugnet_amd import configs, patch_aug_net
from patchaugnet_amd.extract import GraphedExtractor, _pipeline_streams
from patchaugnet_amd.weights import seeded_state_dict, synthetic_submaps
S = int(sys.argv[1]) if len(sys.argv) > 1 else 4
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
    streams = _pipeline_streams(dev, S)
    sides = [torch.cuda.Stream() for _ in range(S)]
    slots = []
    cur = torch.cuda.current_stream()
    for k in range(S):
        st, side = streams[k], sides[k]
        bufs = [(torch.empty(32, m0, dtype=torch.int32, device="cuda"), torch.empty(32, m0, 3, device="cuda")) for _ in range(2)]
        st.wait_stream(cur)
        with torch.cuda.stream(st):
            for b in bufs:
                eng.sample_first_level(xyz, *b)
            for _ in range(2):
                eng.forward(x, views=False, s0=bufs[0])
        cur.wait_stream(st)
        graphs = []
        for a in range(2):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=st, capture_error_mode="thread_local"):
                side.wait_stream(st)
                with torch.cuda.stream(side):
                    eng.sample_first_level(xyz, *bufs[1 - a])
                y, _ = eng.forward(x, views=False, s0=bufs[a])
                st.wait_stream(side)
            graphs.append((g, y))
        slots.append((graphs, st, bufs))
    torch.cuda.synchronize()
    gx = GraphedExtractor(model, tuple(x.shape), S, resident_inputs=[x])

    def run_fork(reps=5):
        rates = []
        phase = [0] * S
        for rep in range(reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for k in range(S):
                slots[k][1].wait_stream(cur)
            for i in range(K):
                k = i % S
                graphs, st, bufs = slots[k]
                if i < S:      # first step of this slot in the region: its samples are not there yet (cold start): sample on the slot's stream first
                    with torch.cuda.stream(st):
                        eng.sample_first_level(xyz, *bufs[phase[k]])
                g, y = graphs[phase[k]]
                with torch.cuda.stream(st):
                    g.replay()
                    descs[i].copy_(y, non_blocking=True)
                phase[k] ^= 1
            for k in range(S):
                cur.wait_stream(slots[k][1])
            torch.cuda.synchronize()
            rates.append(K * 32 / (time.perf_counter() - t0))
        return rates

    def run_base(reps=5):
        rates = []
        for rep in range(reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            gx.begin()
            for i in range(K):
                gx.run(x, out=descs[i])
            gx.end()
            torch.cuda.synchronize()
            rates.append(K * 32 / (time.perf_counter() - t0))
        return rates
    for _ in range(2):
        b = run_base(); f = run_fork()
        ok = all(torch.equal(descs[i], ref) for i in range(K))
        print(f"streams {S} steps {K}: baseline {sorted(b)[len(b)//2]:.0f} ({min(b):.0f}-{max(b):.0f})   forked sampling {sorted(f)[len(f)//2]:.0f} ({min(f):.0f}-{max(f):.0f}) submaps/s  bit-identical {ok}")
