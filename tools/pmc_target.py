#!/usr/bin/env python
"""Workload for the rocprofv3 --pmc passes: 3 engine steps at the bench shape (B=32, 4096 pts) + the K5 micro-benchmark."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from patchaugnet_amd import _lib, configs, patch_aug_net
from patchaugnet_amd.weights import seeded_state_dict, synthetic_submaps

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--points", type=int, default=4096)
ap.add_argument("--model", choices=["patch_aug_net", "pptnet"], default="patch_aug_net")
ap.add_argument("--mlp-dtype", choices=["f32", "f16"], default="f32")
ap.add_argument("--no-grouping", action="store_true")
args = ap.parse_args()
if args.model == "pptnet":
    from patchaugnet_amd import pptnet
    cfg = configs.pptnet_config() if args.points == 4096 else configs.scaled_config(configs.pptnet_config(), args.points)
    model = pptnet.Network(param=cfg, use_normalize=True)
else:
    cfg = configs.patch_aug_net_config() if args.points == 4096 else configs.scaled_config(configs.patch_aug_net_config(), args.points)
    model = patch_aug_net.Network(param=cfg, use_a2a_recon=True, use_l2_norm=True)
model.load_state_dict(seeded_state_dict(model.state_dict()))
model = model.cuda().eval()
model.mlp_dtype = args.mlp_dtype
x = synthetic_submaps(args.batch, args.points, seed=1234).cuda()
with torch.no_grad():
    for _ in range(3):
        model(x, return_feat=False)
if args.no_grouping:
    torch.cuda.synchronize()
    sys.exit(0)
b, c, n, m, k = 4096, 64, 1024, 128, 20
pts = torch.randn(b, c, n, device="cuda")
idx = torch.randint(0, n, (b, m, k), device="cuda", dtype=torch.int32)
o = torch.empty(b, c, m, k, device="cuda")
for _ in range(3):
    _lib.call("pa_grouping_forward", b, c, n, m, k, _lib.ptr(pts), _lib.ptr(idx), _lib.ptr(o))
torch.cuda.synchronize()
