"""train.GraphedTrainer(prefetch=True) at the full training shape (18 x 4096 points), learning rate 0, the same batch announced and trained every
step, eager work between steps: every odd step's parameter gradients (p.grad references the tensors of the buffer set captured last) must equal the first odd step's.
python tools/probes/graph_replay_gradients_prefetch.py [steps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from patchaugnet_amd import configs, patch_aug_net
from patchaugnet_amd.train import DEFAULTS, GraphedTrainer
from patchaugnet_amd.weights import seeded_state_dict
n = 4096
m = patch_aug_net.Network(param=configs.patch_aug_net_config(), use_a2a_recon=True, use_l2_norm=True)
m.load_state_dict(seeded_state_dict(m.state_dict())); m = m.cuda()
g = torch.Generator().manual_seed(5)
batch = tuple((torch.rand(1, k, n, 3, generator=g) * 2 - 1).cuda() for k in (1, 2, 14, 1))
nn_dict = {(0, 1): None, (0, 2): None}
opt = torch.optim.SGD(m.parameters(), lr=0.0)
tr = GraphedTrainer(m, opt, *batch, nn_dict, num_points=n, args=DEFAULTS, warmup=2, prefetch=True)
scratch = torch.empty(1 << 20, device="cuda")
ref = None
worst = 0.0
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
for i in range(steps):
    torch.manual_seed(1)
    tr.step(*batch, next_batch=batch)
    torch.cuda.synchronize()
    cur = {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}
    if i % 2 == 0:
        pass            # p.grad references the gradient tensors of the buffer set captured LAST (set 1): they are written by the odd steps
    elif ref is None:
        ref = cur
    else:
        for k in ref:
            a, b = ref[k], cur[k]
            if not torch.isfinite(b).all():
                print("step", i, "NON-FINITE gradient in", k); worst = float("inf")
            else:
                r = (a - b).norm().item() / max(a.norm().item(), 1e-3)
                if r > 1e-2 and i <= 3: print(f'   step {i} {k}: |ref| {a.norm().item():.3e} |cur| {b.norm().item():.3e} |diff| {(a - b).norm().item():.3e}')
                worst = max(worst, r)
    scratch.fill_(float(i)); junk = [torch.full((1 << 20,), float("nan"), device="cuda") for _ in range(8)]; float(scratch[0]); del junk
print(f"{steps} steps, worst relative gradient difference to the first step: {worst:.3e}", {k: round(float(v), 5) for k, v in tr.losses.items()})
