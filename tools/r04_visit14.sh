#!/bin/bash
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_extract.py -m gpu -q 2>&1 | tail -3
python - <<'PY'
import torch, time
from patchaugnet_amd import configs, patch_aug_net
from patchaugnet_amd.weights import seeded_state_dict, synthetic_submaps
m = patch_aug_net.Network(param=configs.patch_aug_net_config(), use_a2a_recon=True, use_l2_norm=True)
m.load_state_dict(seeded_state_dict(m.state_dict())); m = m.cuda().eval()
import os
def lat(B, mode):
    x = synthetic_submaps(B, 4096, seed=3).cuda()
    m.geo_overlap = mode != "plain"
    if mode == "overlap_nochunk": os.environ["PA_ENGINE_NO_FPS_CHUNKS"] = "1"
    else: os.environ.pop("PA_ENGINE_NO_FPS_CHUNKS", None)
    with torch.no_grad():
        for _ in range(5): m(x, return_feat=False)
        torch.cuda.synchronize(); ts = []
        for _ in range(30):
            t0 = time.perf_counter(); m(x, return_feat=False); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    ts.sort(); return ts[len(ts)//2] * 1e3
import itertools
for B in (1, 8, 32):
    r = {k: round(lat(B, k), 3) for k in ("plain", "overlap_nochunk", "overlap_chunks")}
    for cuts in ("512", "896", "512,768,896"):
        os.environ["PA_ENGINE_FPS_CHUNKS"] = cuts; r["cuts " + cuts] = round(lat(B, "overlap_chunks"), 3); os.environ.pop("PA_ENGINE_FPS_CHUNKS")
    print("B", B, r)
PY
