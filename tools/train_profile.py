#!/usr/bin/env python
"""One config-4 training step repeated a few times (target for rocprofv3 --kernel-trace --stats)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from patchaugnet_amd.hostcpu import limit_host_threads
limit_host_threads()
from patchaugnet_amd import configs, patch_aug_net
from patchaugnet_amd.train import training_step
from patchaugnet_amd.weights import seeded_state_dict
model = patch_aug_net.Network(param=configs.patch_aug_net_config(), use_a2a_recon=True, use_l2_norm=True)
model.load_state_dict(seeded_state_dict(model.state_dict())); model = model.cuda()
g = torch.Generator().manual_seed(5)
q = torch.rand(1, 1, 4096, 3, generator=g) * 2 - 1
pos = torch.rand(1, 2, 4096, 3, generator=g) * 2 - 1
neg = torch.rand(1, 14, 4096, 3, generator=g) * 2 - 1
oth = torch.rand(1, 1, 4096, 3, generator=g) * 2 - 1
nn_dict = {(0, 1): torch.randint(0, 4096, (1024, 1), generator=g).numpy(), (0, 2): torch.randint(0, 4096, (1024, 1), generator=g).numpy()}
opt = torch.optim.Adam(model.parameters(), lr=1e-5)
for i in range(int(os.environ.get("STEPS", "6"))):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    training_step(model, opt, q, pos, neg, oth, nn_dict=nn_dict)
    torch.cuda.synchronize(); print(f"step {i}: {(time.perf_counter()-t0)*1e3:.2f} ms", flush=True)
