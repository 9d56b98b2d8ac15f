import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from patchaugnet_amd.hostcpu import limit_host_threads
limit_host_threads()
from patchaugnet_amd import configs, patch_aug_net
from patchaugnet_amd.extract import StreamPipeline
from patchaugnet_amd.weights import seeded_state_dict, synthetic_submaps
m = patch_aug_net.Network(param=configs.patch_aug_net_config(), use_a2a_recon=True, use_l2_norm=True)
m.load_state_dict(seeded_state_dict(m.state_dict())); m = m.cuda().eval()
x = synthetic_submaps(1, 4096, seed=1).cuda()
with torch.no_grad():
    pipe = StreamPipeline(12)
    pipe.begin()
    for _ in range(24): pipe.submit(lambda: m(x, return_feat=False))
    pipe.end(); torch.cuda.synchronize()
    pr = cProfile.Profile(); pr.enable()
    pipe.begin()
    for _ in range(300): pipe.submit(lambda: m(x, return_feat=False))
    pipe.end(); torch.cuda.synchronize()
    pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
