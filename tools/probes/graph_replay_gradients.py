"""Two replays of train.GraphedTrainer's captured step at learning rate 0 must give the same parameter gradients; argument "fill" runs an eager launch
between them.  This is the probe that found the memset-node ordering problem of round 5 (hipMemsetAsync in pa_chamfer_backward captured into the graph:
DESIGN.md section 5); tests/test_gpu_train_ops.py::test_graphed_training_step_with_eager_launches_between_replays is its pytest form.
python tools/probes/graph_replay_gradients.py none|fill"""
import copy, sys, os
sys.path.insert(0, os.environ.get("DBG_ROOT") or os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from patchaugnet_amd import configs, patch_aug_net
from patchaugnet_amd.train import DEFAULTS, GraphedTrainer
from patchaugnet_amd.weights import seeded_state_dict
n, negs = 1024, 4
cfg = configs.scaled_config(configs.patch_aug_net_config(), n)
m = patch_aug_net.Network(param=cfg, use_a2a_recon=True, use_l2_norm=True)
m.load_state_dict(seeded_state_dict(m.state_dict())); m = m.cuda()
g = torch.Generator().manual_seed(5)
batch = tuple((torch.rand(1, k, n, 3, generator=g) * 2 - 1).cuda() for k in (1, 2, negs, 1))
nn_dict = {(0, 1): None, (0, 2): None}
args = dict(DEFAULTS, TRAIN_NEGATIVES_PER_QUERY=negs)
opt = torch.optim.SGD(m.parameters(), lr=0.0)             # no weight change: two replays must give the same gradients
pre = torch.empty(1 << 20, device="cuda")
tr = GraphedTrainer(m, opt, *batch, nn_dict, num_points=n, args=args, warmup=2)
torch.manual_seed(1); tr.step(*batch); torch.cuda.synchronize()
ga = {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}
la = float(tr.losses["total"])
if sys.argv[1] == "fill": pre.fill_(3.0); torch.cuda.synchronize()
torch.manual_seed(1); tr.step(*batch); torch.cuda.synchronize()
lb = float(tr.losses["total"])
print(sys.argv[1], "losses", la, lb)
names = [k for k, _ in m.named_parameters() if k in ga]
for k in reversed(names):
    a, b = ga[k], dict(m.named_parameters())[k].grad
    na, nb = float(a.norm()), float(b.norm())
    d = float((a - b).norm())
    if na > 0 or nb > 0:
        flag = "" if d <= 1e-2 * max(na, 1e-12) else "   <<<<"
        print(f"  {k:58s} |a|={na:.3e} |b|={nb:.3e} |a-b|={d:.3e}{flag}")
