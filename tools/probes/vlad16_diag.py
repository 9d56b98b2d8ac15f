import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from patchaugnet_amd import configs, pptnet, patch_aug_net, engine
from tests._util import golden, seeded_sd_from_table
orig = engine._Pyramid.launch
def launch(self, state, feats, out, phases):
    for v, f in zip(self.vlads, feats):
        if f is not None and v.k == 64:
            w = v.wc_t
            lg = f.reshape(-1, 256) @ w
            print("feat absmax %.2f mean|x| %.3f  |W| max %.3f  logits absmax %.2f std %.2f" % (f.abs().max().item(), f.abs().mean().item(), w.abs().max().item(), lg.abs().max().item(), lg.std().item()))
    return orig(self, state, feats, out, phases)
engine._Pyramid.launch = launch
for name in ("pptnet", "patch_aug_net"):
    g = golden(name)
    for tag in ("small", "full"):
        if name == "pptnet":
            cfg = configs.pptnet_config()
            if tag == "small": cfg = configs.scaled_config(cfg, 1024)
            m = pptnet.Network(param=cfg, use_normalize=True)
        else:
            cfg = configs.patch_aug_net_config()
            if tag == "small": cfg = configs.scaled_config(cfg, 512)
            m = patch_aug_net.Network(param=cfg, use_a2a_recon=True, use_l2_norm=True)
        m.load_state_dict(seeded_sd_from_table(name), strict=True)
        m = m.cuda().eval(); m.mlp_dtype = "f16"
        print(name, tag)
        with torch.no_grad(): m(torch.from_numpy(g[f"{tag}_x"]).cuda())
