"""Which lines of the module path trigger real layout copies in one training step?"""
import os, sys, collections, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
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
nn_dict = {(0, 1): None, (0, 2): None}
opt = torch.optim.Adam(model.parameters(), lr=1e-5)
training_step(model, opt, q, pos, neg, oth, nn_dict=nn_dict)
counts = collections.Counter(); bytes_ = collections.Counter()
orig = torch.Tensor.contiguous
def patched(self, *a, **k):
    if self.is_cuda and not self.is_contiguous():
        fr = [f for f in traceback.extract_stack()[:-1] if "patchaugnet_amd" in f.filename][-1:]
        key = f"{os.path.basename(fr[0].filename)}:{fr[0].lineno} {fr[0].line}" if fr else "?"
        counts[key] += 1; bytes_[key] += self.numel() * self.element_size()
    return orig(self, *a, **k)
torch.Tensor.contiguous = patched
training_step(model, opt, q, pos, neg, oth, nn_dict=nn_dict)
torch.Tensor.contiguous = orig
for k, c in counts.most_common(25):
    print(f"{c:3d} x {bytes_[k]/c/1e6:8.2f} MB  {k}")
print("forward-side .contiguous() copies:", sum(counts.values()), " total MB", sum(bytes_.values()) / 1e6)
