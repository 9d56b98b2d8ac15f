import os, sys, time, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from patchaugnet_amd.hostcpu import limit_host_threads
limit_host_threads()
from patchaugnet_amd import configs, patch_aug_net, losses
from patchaugnet_amd.train import run_model, DEFAULTS as args
from patchaugnet_amd.weights import seeded_state_dict
model = patch_aug_net.Network(param=configs.patch_aug_net_config(), use_a2a_recon=True, use_l2_norm=True)
model.load_state_dict(seeded_state_dict(model.state_dict())); model = model.cuda().train()
g = torch.Generator().manual_seed(5)
q = torch.rand(1, 1, 4096, 3, generator=g) * 2 - 1
pos = torch.rand(1, 2, 4096, 3, generator=g) * 2 - 1
neg = torch.rand(1, 14, 4096, 3, generator=g) * 2 - 1
oth = torch.rand(1, 1, 4096, 3, generator=g) * 2 - 1
nn_dict = {(0, 1): torch.randint(0, 4096, (1024, 1), generator=g).numpy(), (0, 2): torch.randint(0, 4096, (1024, 1), generator=g).numpy()}
opt = torch.optim.Adam(model.parameters(), lr=1e-5)
def T():
    torch.cuda.synchronize(); return time.perf_counter()
for i in range(14):
    if i == 8: gc.disable()
    t0 = T(); opt.zero_grad(set_to_none=True)
    out = run_model(model, q, pos, neg, oth, nn_dict, 4096, True, args=args); t1 = T()
    oq, op, on, oo = out["global_desc"]
    l = losses.quadruplet_loss(oq, op, on, oo, 0.5, 0.2, lazy=True)
    r = out["patch_recon"]; l = l + losses.get_loss_func("patch_chamfer")(r["origin_patches"], r["reconstructed_patches"]); t2 = T()
    l.backward(); t3 = T()
    opt.step(); t4 = T()
    print(f"step {i}: fwd {1e3*(t1-t0):.1f} loss {1e3*(t2-t1):.1f} bwd {1e3*(t3-t2):.1f} opt {1e3*(t4-t3):.1f} total {1e3*(t4-t0):.1f}  mem {torch.cuda.memory_reserved()>>20} MiB", flush=True)
