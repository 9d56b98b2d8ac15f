#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
./tools/probes/l2_hotspot.bin
python - <<'PY'
import os, numpy as np, torch
from patchaugnet_amd import configs, pptnet
from tests._util import golden, seeded_sd_from_table
g = golden("pptnet")
def cos(a, b): return (a * b).sum(1) / (np.linalg.norm(a, axis=1) * np.linalg.norm(b, axis=1))
for env in ({}, {"PA_ATTN_F16_SPLIT": "0"}, {"PA_ATTN_F16_QV": "1"}, {"PA_ATTN_F16_SPLIT": "0", "PA_ATTN_F16_QV": "1"}):
    for k in ("PA_ATTN_F16_SPLIT", "PA_ATTN_F16_QV"): os.environ.pop(k, None)
    os.environ.update(env)
    for tag, npts in (("small", 1024), ("full", 4096)):
        cfg = configs.pptnet_config() if tag == "full" else configs.scaled_config(configs.pptnet_config(), 1024)
        m = pptnet.Network(param=cfg, use_normalize=True)
        m.load_state_dict(seeded_sd_from_table("pptnet"), strict=True)
        m = m.cuda().eval(); m.mlp_dtype = "f16"
        with torch.no_grad():
            d, _, _ = m(torch.from_numpy(g[f"{tag}_x"]).cuda())
            m.mlp_dtype = "f32"
            d32, _, _ = m(torch.from_numpy(g[f"{tag}_x"]).cuda())
        print(env, tag, "cos f16 vs ref", cos(d.cpu().numpy(), g[f"{tag}_desc_l2"]), "f32 vs ref maxabs", np.abs(d32.cpu().numpy() - g[f"{tag}_desc_l2"]).max())
PY
