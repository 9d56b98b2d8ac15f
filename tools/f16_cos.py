"""cosine of the fp16-path descriptors against the golden fp32 descriptors (both models, small / full fixtures); env knobs select variants"""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from patchaugnet_amd import configs, patch_aug_net, pptnet
from tests._util import golden, seeded_sd_from_table


def cos(a, b):
    return (a * b).sum(1) / (np.linalg.norm(a, axis=1) * np.linalg.norm(b, axis=1))


for name in ("pptnet", "patch_aug_net"):
    g = golden(name)
    for tag in ("small", "full"):
        if name == "pptnet":
            cfg = configs.pptnet_config()
            if tag == "small":
                cfg = configs.scaled_config(cfg, 1024)
            m = pptnet.Network(param=cfg, use_normalize=True)
            ref = g[f"{tag}_desc_l2"]
        else:
            cfg = configs.patch_aug_net_config()
            if tag == "small":
                cfg = configs.scaled_config(cfg, 512)
            m = patch_aug_net.Network(param=cfg, use_a2a_recon=True, use_l2_norm=True)
            ref = g[f"{tag}_desc"]
        m.load_state_dict(seeded_sd_from_table(name), strict=True)
        m = m.cuda().eval()
        m.mlp_dtype = "f16"
        with torch.no_grad():
            d = m(torch.from_numpy(g[f"{tag}_x"]).cuda())[0].cpu().numpy()
        c = cos(d, ref)
        print(f"{name} {tag}: cos min {c.min():.6f} mean {c.mean():.6f}  max|d| {np.abs(d - ref).max():.2e}", flush=True)
