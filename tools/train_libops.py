#!/usr/bin/env python
"""Which lines of the training step still launch library (at::native / copy / fill) kernels?  Two eager config-3 steps under torch.profiler with stacks:
every aten op with device time, grouped by op and by the innermost frames inside this repository.  usage: python tools/train_libops.py [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from patchaugnet_amd.hostcpu import limit_host_threads
limit_host_threads()
from patchaugnet_amd import configs, patch_aug_net
from patchaugnet_amd.train import training_step
from patchaugnet_amd.weights import seeded_state_dict

STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 2
model = patch_aug_net.Network(param=configs.patch_aug_net_config(), use_a2a_recon=True, use_l2_norm=True)
model.load_state_dict(seeded_state_dict(model.state_dict())); model = model.cuda()
g = torch.Generator().manual_seed(5)
q = torch.rand(1, 1, 4096, 3, generator=g) * 2 - 1
pos = torch.rand(1, 2, 4096, 3, generator=g) * 2 - 1
neg = torch.rand(1, 14, 4096, 3, generator=g) * 2 - 1
oth = torch.rand(1, 1, 4096, 3, generator=g) * 2 - 1
q, pos, neg, oth = (t.cuda() for t in (q, pos, neg, oth))
nn_dict = {(0, 1): torch.randint(0, 4096, (1024, 1), generator=g).numpy(), (0, 2): torch.randint(0, 4096, (1024, 1), generator=g).numpy()}
opt = torch.optim.Adam(model.parameters(), lr=1e-5, capturable=True, fused=True)
for _ in range(3):
    training_step(model, opt, q, pos, neg, oth, nn_dict=nn_dict)
torch.cuda.synchronize()

# ---- every aten op of one step with the innermost frames of this repository (dispatch mode; the backward runs on this thread so that it is seen too)
import traceback
from torch.utils._python_dispatch import TorchDispatchMode
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SKIP = ("aten::view", "aten::_unsafe_view", "aten::empty", "aten::as_strided", "aten::detach", "aten::t", "aten::transpose", "aten::permute", "aten::slice",
        "aten::select", "aten::unsqueeze", "aten::squeeze", "aten::expand", "aten::alias", "aten::reshape", "aten::_reshape_alias", "aten::empty_like",
        "aten::empty_strided", "aten::new_empty", "aten::unbind", "aten::split", "aten::lift_fresh", "aten::is_", "aten::sym_", "aten::result_type",
        "aten::_local_scalar_dense", "aten::item", "aten::new_empty_strided", "aten::stride", "aten::size", "aten::storage_offset", "aten::numel")
log = {}


class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func._schema.name
        out = func(*args, **(kwargs or {}))
        if not name.startswith(SKIP):
            ts = [a for a in args if isinstance(a, torch.Tensor)]
            if any(t.is_cuda for t in ts) or (isinstance(out, torch.Tensor) and out.is_cuda):
                fr = [f for f in traceback.extract_stack() if root in f.filename and "train_libops" not in f.filename][-3:]
                where = " <- ".join(f"{os.path.relpath(f.filename, root)}:{f.lineno}" for f in reversed(fr)) or "(no repo frame: built-in backward node)"
                shp = ",".join("x".join(map(str, t.shape)) for t in ts[:3])
                key = (name, where, shp)
                log[key] = log.get(key, 0) + 1
        return out


from patchaugnet_amd import losses, train_ops
from patchaugnet_amd.train import run_model, DEFAULTS as A


def body(geo):          # GraphedTrainer._body
    with train_ops.zero_arena(torch.device("cuda", 0)):
        out = run_model(model, q, pos, neg, oth, nn_dict, 4096, True, args=A, geometry=geo)
        oq, op, on, oo = out["global_desc"]
        total = losses.quadruplet_loss(oq, op, on, oo, A["MARGIN_1"], A["MARGIN_2"], use_min=A["TRIPLET_USE_BEST_POSITIVES"], lazy=A["LOSS_LAZY"],
                                       ignore_zero_loss=A["LOSS_IGNORE_ZERO_BATCH"])
        r = out["patch_recon"]
        total = total + losses.patch_chamfer_loss(r["origin_patches"], r["reconstructed_patches"])
        total.backward()
    opt.step()


torch.autograd.set_multithreading_enabled(False)
feed = torch.cat([q, pos, neg, oth], 1).view(-1, 1, 4096, 3)
with Log():
    geo = model.backbone.geometry(feed.squeeze(1))
print(f"---- geometry (prefetch graph, side stream): {sum(log.values())} device aten ops")
for (name, where, shp), c in sorted(log.items(), key=lambda kv: (-kv[1], kv[0])):
    print(f"{c:4d}x {name:24s} [{shp}]  {where}")
log.clear()
opt.zero_grad(set_to_none=True)
with Log():
    body(geo)
print("---- step graph")
torch.cuda.synchronize()
print(f"{sum(log.values())} device aten ops in one step (views / allocations not counted)")
for (name, where, shp), c in sorted(log.items(), key=lambda kv: (-kv[1], kv[0])):
    print(f"{c:4d}x {name:24s} [{shp}]  {where}")
sys.exit(0)
