"""Which Python lines of the training step launch library (at::native) kernels: one eager step under torch.profiler with stacks, device
kernels grouped by the innermost patchaugnet_amd frame of the launching op.  python tools/probes/train_glue.py"""
import collections
import os
import sys
import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from patchaugnet_amd import configs, patch_aug_net
from patchaugnet_amd.train import training_step
from patchaugnet_amd.weights import seeded_state_dict

cfg = configs.patch_aug_net_config()
model = patch_aug_net.Network(param=cfg, use_a2a_recon=True, use_l2_norm=True)
model.load_state_dict(seeded_state_dict(model.state_dict()))
model = model.cuda()
g = torch.Generator().manual_seed(5)
n = 4096
q, pos, neg, oth = ((torch.rand(1, k, n, 3, generator=g) * 2 - 1).cuda() for k in (1, 2, 14, 1))
nn_dict = {(0, 1): torch.randint(0, n, (1024, 1), generator=g).numpy(), (0, 2): torch.randint(0, n, (1024, 1), generator=g).numpy()}
opt = torch.optim.Adam(model.parameters(), lr=1e-5, fused=True)
step = lambda: training_step(model, opt, q, pos, neg, oth, nn_dict=nn_dict, num_points=n)
for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
by_line = collections.defaultdict(lambda: [0, 0.0, collections.Counter()])
for ev in prof.events():
    if ev.device_type != torch.autograd.DeviceType.CPU or not ev.kernels:
        continue
    native = [k for k in ev.kernels if "anonymous namespace)::" not in k.name or "at::native" in k.name]
    if not native:
        continue
    frame = next((s for s in ev.stack if "patchaugnet_amd/" in s), None)
    if frame is None:
        frame = "backward/other: " + ev.name
    else:
        frame = frame.split("patchaugnet_amd/")[-1]
    rec = by_line[frame]
    rec[0] += len(native)
    rec[1] += sum(k.duration for k in native)
    rec[2][ev.name] += len(native)
tot_n = sum(r[0] for r in by_line.values())
tot_t = sum(r[1] for r in by_line.values())
print(f"library kernels in one step: {tot_n} launches, {tot_t:.0f} us")
for frame, (cnt, us, ops) in sorted(by_line.items(), key=lambda kv: -kv[1][1])[:60]:
    print(f"{cnt:4d} {us:8.1f} us  {frame[:90]:90s} {dict(ops.most_common(4))}")
