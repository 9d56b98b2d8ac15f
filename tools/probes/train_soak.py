"""Soak of the captured training step at the full shape: train.GraphedTrainer(prefetch=True) + patchaugnet_amd.optim.Adam with a StepLR schedule, a NEW
random tuple every step (fresh device allocations between replays), a few hundred steps.  Losses, weights and optimizer state must stay finite and the
reconstruction loss must fall.  python tools/probes/train_soak.py [steps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from patchaugnet_amd import configs, patch_aug_net
from patchaugnet_amd.optim import Adam
from patchaugnet_amd.train import DEFAULTS, GraphedTrainer
from patchaugnet_amd.weights import seeded_state_dict
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
n = 4096
m = patch_aug_net.Network(param=configs.patch_aug_net_config(), use_a2a_recon=True, use_l2_norm=True)
m.load_state_dict(seeded_state_dict(m.state_dict())); m = m.cuda()
g = torch.Generator().manual_seed(7)
def tuple_():
    base = torch.rand(1, 1, n, 3, generator=g) * 2 - 1
    q = base + 0.01 * torch.randn(1, 1, n, 3, generator=g)
    pos = base + 0.02 * torch.randn(1, 2, n, 3, generator=g)
    neg = torch.rand(1, 14, n, 3, generator=g) * 2 - 1
    oth = torch.rand(1, 1, n, 3, generator=g) * 2 - 1
    return tuple(t.cuda() for t in (q, pos, neg, oth))
nn_dict = {(0, 1): None, (0, 2): None}
opt = Adam(m.parameters(), lr=1e-4)
sched = torch.optim.lr_scheduler.StepLR(opt, step_size=100, gamma=0.5)
cur = tuple_()
tr = GraphedTrainer(m, opt, *cur, nn_dict, num_points=n, args=DEFAULTS, warmup=2, prefetch=True)
hist = []
for i in range(steps):
    nxt = tuple_()
    l = tr.step(*cur, next_batch=nxt)
    if i % 25 == 0 or i == steps - 1:
        hist.append((i, round(float(l["place_recognition"]), 4), round(float(l["patch_recon_a2a"]), 4), opt.param_groups[0]["lr"]))
    cur = nxt
    sched.step()
torch.cuda.synchronize()
okw = all(bool(torch.isfinite(v).all()) for v in m.state_dict().values() if v.is_floating_point())
oks = all(bool(torch.isfinite(t).all()) for st in opt.state.values() for t in st.values() if torch.is_tensor(t))
vmax = max(float(st["exp_avg_sq"].max()) for st in opt.state.values())
print("history (step, place, recon, lr):", hist)
print("weights finite:", okw, " optimizer state finite:", oks, " max exp_avg_sq:", f"{vmax:.3e}", " device lr:", float(opt._lr[0][0]))
