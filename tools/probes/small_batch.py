import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from patchaugnet_amd.hostcpu import limit_host_threads
limit_host_threads()
from patchaugnet_amd import configs, patch_aug_net
from patchaugnet_amd.extract import StreamPipeline
from patchaugnet_amd.weights import seeded_state_dict, synthetic_submaps
m = patch_aug_net.Network(param=configs.patch_aug_net_config(), use_a2a_recon=True, use_l2_norm=True)
m.load_state_dict(seeded_state_dict(m.state_dict())); m = m.cuda().eval()
def rate(B, S, K):
    x = synthetic_submaps(B, 4096, seed=1).cuda()
    out = torch.empty(K, B, 256, device="cuda")
    pipe = StreamPipeline(S)
    def one(i):
        d = m(x, return_feat=False)
        if i >= 0: out[i].copy_(d)
    with torch.no_grad():
        pipe.begin()
        for _ in range(2 * S): pipe.submit(one, -1)
        pipe.end(); torch.cuda.synchronize()
        t0 = time.perf_counter(); pipe.begin()
        for i in range(K): pipe.submit(one, i)
        pipe.end(); t_host = time.perf_counter() - t0
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    return K * B / dt, dt / K * 1e3, t_host / K * 1e3
import itertools
keep = []
def parse_case(c):
    b, s_ = c.split(":")
    return (b, int(s_)) if b == "dummy" else (int(b), int(s_))
cases = [parse_case(c) for c in sys.argv[1:]] if len(sys.argv) > 1 else list(itertools.product((1, 8, 32), (1, 3, 6, 12)))
for B, S in cases:
    if B == "dummy":      # activate S more pool streams with a trivial kernel each
        for _ in range(S):
            st = torch.cuda.Stream(); keep.append(st)
            with torch.cuda.stream(st):
                torch.zeros(8, device="cuda").add_(1)
        torch.cuda.synchronize(); print("activated", S, "dummy streams"); continue
    if True:
        r, ms, host = rate(B, S, 120)
        print(f"B={B:3d} streams={S:2d}: {r:8.0f} submaps/s  {ms:.3f} ms/step  host submit {host:.3f} ms/step", flush=True)
