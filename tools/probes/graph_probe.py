import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from patchaugnet_amd.hostcpu import limit_host_threads
limit_host_threads()
from patchaugnet_amd import configs, patch_aug_net
from patchaugnet_amd.weights import seeded_state_dict, synthetic_submaps
m = patch_aug_net.Network(param=configs.patch_aug_net_config(), use_a2a_recon=True, use_l2_norm=True)
m.load_state_dict(seeded_state_dict(m.state_dict())); m = m.cuda().eval()
for B in (1, 8, 32):
    x = synthetic_submaps(B, 4096, seed=1).cuda()
    with torch.no_grad():
        ref = m(x, return_feat=False).clone()
        S = 4
        graphs = []
        for s in range(S):
            xs = x.clone()
            st = torch.cuda.Stream()
            st.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(st):
                for _ in range(2): m(xs, return_feat=False)
            torch.cuda.current_stream().wait_stream(st)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=st):
                ys = m(xs, return_feat=False)
            graphs.append((g, xs, ys, st))
        torch.cuda.synchronize()
        for g, xs, ys, st in graphs:
            with torch.cuda.stream(st): g.replay()
        torch.cuda.synchronize()
        print("B", B, "graph == eager:", all(torch.equal(ys, ref) for _, _, ys, _ in graphs))
        K = 200
        t0 = time.perf_counter()
        for i in range(K):
            g, xs, ys, st = graphs[i % S]
            with torch.cuda.stream(st): g.replay()
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"B={B} graphs on {S} streams: {K*B/dt:.0f} submaps/s  {dt/K*1e3:.3f} ms/step  host {t_host/K*1e3:.3f} ms/step")
