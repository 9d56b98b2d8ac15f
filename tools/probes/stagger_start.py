"""Probe: break the lock-step of the four pipeline streams at the START of a region.  The first step of stream k is issued as [eager first-level
sampling -> event E_k -> graph of the rest], and stream k + 1's first sampling waits for E_k: the four sampling chains of the first round run one
after another (under the previous streams' dense kernels) instead of all at once with the chip idle.  python tools/probes/stagger_start.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from patchaugnet_amd import configs, patch_aug_net
from patchaugnet_amd.extract import GraphedExtractor
from patchaugnet_amd.weights import seeded_state_dict, synthetic_submaps
S = 4
K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
MODE = sys.argv[2] if len(sys.argv) > 2 else "chain"
model = patch_aug_net.Network(param=configs.patch_aug_net_config(), use_a2a_recon=True, use_l2_norm=True)
model.load_state_dict(seeded_state_dict(model.state_dict()))
model = model.cuda().eval()
x = synthetic_submaps(32, 4096, seed=1234).cuda()
xyz = x.squeeze(1).contiguous()
descs = torch.empty(K, 32, 256, device="cuda")
with torch.no_grad():
    ref = model(x, return_feat=False).clone()
    gx = GraphedExtractor(model, tuple(x.shape), S, resident_inputs=[x])
    eng = model._engine
    m0 = eng.sampling[0]
    cur = torch.cuda.current_stream()
    rest = []
    for k in range(S):
        st = gx.slots[k][3]
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
        rest.append((g, y, buf))
    torch.cuda.synchronize()

    def region(stagger):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        gx.begin()
        prev = None
        for i in range(K):
            k = i % S
            g_full, xs, ys, st = gx.slots[k]
            with torch.cuda.stream(st):
                if stagger and i < S:
                    if prev is not None and (MODE == "chain" or (MODE == "pairs" and k == 2)):
                        st.wait_event(prev)
                    eng.sample_first_level(xyz, *rest[k][2])
                    ev = torch.cuda.Event()
                    ev.record(st)
                    if MODE == "chain" or (MODE == "pairs" and k == 1):
                        prev = ev
                    rest[k][0].replay()
                    descs[i].copy_(rest[k][1], non_blocking=True)
                else:
                    g_full.replay()
                    descs[i].copy_(ys, non_blocking=True)
        gx.end()
        torch.cuda.synchronize()
        return K * 32 / (time.perf_counter() - t0)
    for _ in range(3):
        b = sorted(region(False) for _ in range(5)); s = sorted(region(True) for _ in range(5))
        ok = all(torch.equal(descs[i], ref) for i in range(K))
        print(f"steps {K} mode {MODE}: lock-step start {b[2]:.0f} ({b[0]:.0f}-{b[-1]:.0f})   staggered start {s[2]:.0f} ({s[0]:.0f}-{s[-1]:.0f}) submaps/s  bit-identical {ok}")
