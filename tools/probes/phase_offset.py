"""Probe: start the pipeline streams of a region with phase offsets (a spin kernel of k x D microseconds in front of stream k's first replay) instead of
in lock-step.  python tools/probes/phase_offset.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from patchaugnet_amd import configs, patch_aug_net
from patchaugnet_amd.extract import GraphedExtractor
from patchaugnet_amd.weights import seeded_state_dict, synthetic_submaps
K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
model = patch_aug_net.Network(param=configs.patch_aug_net_config(), use_a2a_recon=True, use_l2_norm=True)
model.load_state_dict(seeded_state_dict(model.state_dict()))
model = model.cuda().eval()
x = synthetic_submaps(32, 4096, seed=1234).cuda()
descs = torch.empty(K, 32, 256, device="cuda")
CYC_PER_US = 100          # torch.cuda._sleep counts clock64 ticks (100 MHz wall clock on gfx9)
with torch.no_grad():
    gx = GraphedExtractor(model, tuple(x.shape), 4, resident_inputs=[x])
    def region(offsets_us):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        gx.begin()
        for i in range(K):
            k = gx._i % 4
            if i < 4 and offsets_us[k] > 0:
                with torch.cuda.stream(gx.slots[k][3]):
                    torch.cuda._sleep(int(offsets_us[k] * CYC_PER_US))
            gx.run(x, out=descs[i])
        gx.end()
        torch.cuda.synchronize()
        return K * 32 / (time.perf_counter() - t0)
    # calibrate the sleep
    torch.cuda.synchronize(); t0 = time.perf_counter(); torch.cuda._sleep(100 * 1000); torch.cuda.synchronize(); print(f"_sleep(100000) = {(time.perf_counter() - t0) * 1e6:.0f} us")
    pats = {"lock-step": (0, 0, 0, 0), "0/150/300/450": (0, 150, 300, 450), "0/250/500/750": (0, 250, 500, 750), "0/0/400/400": (0, 0, 400, 400), "0/400/0/400": (0, 400, 0, 400),
            "0/100/200/300": (0, 100, 200, 300), "0/0/0/500": (0, 0, 0, 500)}
    for rep in range(2):
        for name, off in pats.items():
            r = sorted(region(off) for _ in range(5))
            print(f"steps {K} offsets {name:16s}: {r[2]:.0f} ({r[0]:.0f}-{r[-1]:.0f}) submaps/s")
