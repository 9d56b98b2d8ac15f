"""Graph capture next to an initialised RCCL process group (watchdog thread polling events): world size 1 on one GPU."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import socket
with socket.socket() as _s:                       # a free port: a fixed one can still be in TIME_WAIT from the previous run
    _s.bind(("127.0.0.1", 0)); _port = _s.getsockname()[1]
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", str(_port))
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
t = torch.ones(1024, device="cuda"); dist.all_reduce(t); torch.cuda.synchronize()
from patchaugnet_amd import configs, patch_aug_net
from patchaugnet_amd.extract import GraphedExtractor
from patchaugnet_amd.weights import seeded_state_dict, synthetic_submaps
m = patch_aug_net.Network(param=configs.patch_aug_net_config(), use_a2a_recon=True, use_l2_norm=True)
m.load_state_dict(seeded_state_dict(m.state_dict())); m = m.cuda().eval()
x = synthetic_submaps(32, 4096, seed=1).cuda()
with torch.no_grad():
    ref = m(x, return_feat=False).clone()
    for trial in range(3):
        pending = [dist.all_reduce(t, async_op=True) for _ in range(50)]      # keep the watchdog busy during capture
        gx = GraphedExtractor(m, tuple(x.shape), 4)
        for w in pending: w.wait()
        out = torch.empty(8, 32, 256, device="cuda")
        gx.begin()
        for i in range(8): gx.run(x, out=out[i])
        gx.end()
        g = torch.empty(8 * 32, 256, device="cuda")
        dist.all_gather_into_tensor(g, out.view(-1, 256))
        torch.cuda.synchronize()
        print("trial", trial, "ok", all(torch.equal(out[i], ref) for i in range(8)), torch.equal(g, out.view(-1, 256)))
dist.destroy_process_group()
