import sys, os
sys.path.insert(0, os.getcwd())
import torch
from patchaugnet_amd import train_ops, loupe
torch.manual_seed(0)
for b, ktot in [(1, 93), (2, 93), (1, 84), (1, 96), (3, 7), (1, 1), (5, 129)]:
    x = torch.randn(b, 256, ktot, device="cuda")
    W = torch.randn(256, 256, 1, device="cuda") * 0.05
    y = train_ops.linear_cm(x, W)
    ref = torch.matmul(W.squeeze(-1).double(), x.double())
    e1 = (y.double() - ref).abs().max().item()
    Wf = torch.randn(256, 256 * ktot, device="cuda") * 0.01
    bias = torch.randn(256, device="cuda")
    xf = x.flatten(1).contiguous()
    z = train_ops.linear_rows(xf, Wf, bias)
    refz = xf.double() @ Wf.double().t() + bias.double()
    e2 = (z.double() - refz).abs().max().item()
    g = train_ops.matmul_rows(xf[:, :256].contiguous(), W.squeeze(-1).contiguous())
    e3 = (g.double() - xf[:, :256].double() @ W.squeeze(-1).double()).abs().max().item()
    print(b, ktot, "linear_cm err %.2e  linear_rows err %.2e  matmul_rows err %.2e" % (e1, e2, e3))
