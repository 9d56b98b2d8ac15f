#!/usr/bin/env python
"""Warm back-to-back duration of every chain launch of a step: each chain call is repeated R times in place (idempotent: outputs are
overwritten); (t_R - t_1) / (R - 1) is the launch's duration with its weights and gather sources already in L2, free of event overhead.
Next to the single-launch stage time it tells how much of a stage is cold-cache / launch latency.  python tools/probes/stage_b2b.py [R]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from patchaugnet_amd import configs, patch_aug_net, profiling
from patchaugnet_amd.engine import _Chain
from patchaugnet_amd.weights import seeded_state_dict, synthetic_submaps

R = int(sys.argv[1]) if len(sys.argv) > 1 else 9
model = patch_aug_net.Network(param=configs.patch_aug_net_config(), use_a2a_recon=True, use_l2_norm=True)
model.load_state_dict(seeded_state_dict(model.state_dict()))
model = model.cuda().eval()
x = synthetic_submaps(32, 4096, seed=1234).cuda()
rep = {"n": 1}
for nm in ("sa", "fp", "fp_premul"):
    orig = getattr(_Chain, nm)

    def wrapped(self, *a, _o=orig, **k):
        r = _o(self, *a, **k)
        for _ in range(rep["n"] - 1):
            r = _o(self, *a, **k)
        return r
    setattr(_Chain, nm, wrapped)
with torch.no_grad():
    one = profiling.engine_stage_times(model, x, 5)
    rep["n"] = R
    many = profiling.engine_stage_times(model, x, 5)
for k in one:
    if "chain" in k or "premul" in k:
        print(f"{k:12s} single {one[k] * 1e3:7.1f} us   warm back-to-back {(many[k] - one[k]) / (R - 1) * 1e3:7.1f} us")
