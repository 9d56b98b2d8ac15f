"""Launch time of the second set-abstraction level's chain (sa1) with and without the LDS-resident kernel (sa_mid.hip).
usage: python tools/sa1_time.py    (MI355X; us per launch: HIP events around 50 back-to-back launches, best of 5)"""
import sys
import torch
sys.path.insert(0, ".")
from patchaugnet_amd import _lib
from patchaugnet_amd.engine import _Chain
from tests.test_gpu_chain import make_layers, sa_inputs

for name, B, n, m, n2 in (("PatchAugNet sa1", 32, 1024, 128, 256), ("PPT-Net sa1", 32, 1024, 256, 128)):
    _, eng = make_layers([67, 64, 64, n2], seed=1)
    xyz, feat, cidx, nbr = sa_inputs(B, n, m, 20, 64, seed=2)
    args = (xyz.cuda(), feat.cuda().contiguous(), cidx.cuda(), nbr.cuda(), 64)
    ch = _Chain(eng)
    for mode in (0, 1):
        _lib.lib().pa_chain_mid_enable(mode)
        best = 1e9
        for rep in range(5):
            ch.sa(*args, pooled=True)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                ch.sa(*args, pooled=True)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 50 * 1e3)
        print(f"{name}: {'sa_mid' if mode else 'generic pooled'} {best:.1f} us per launch", flush=True)
    _lib.lib().pa_chain_mid_enable(-1)
