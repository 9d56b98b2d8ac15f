#!/usr/bin/env python
"""Every GEMM launch of one training step (BASELINE configs[3]) with its shape, duration (HIP events), TFLOP/s and GB/s: where the dense
time of the step goes.  python tools/train_gemm_shapes.py"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from patchaugnet_amd.hostcpu import limit_host_threads
limit_host_threads()
from patchaugnet_amd import configs, patch_aug_net, train_ops
from patchaugnet_amd.train import training_step
from patchaugnet_amd.weights import seeded_state_dict
model = patch_aug_net.Network(param=configs.patch_aug_net_config(), use_a2a_recon=True, use_l2_norm=True)
model.load_state_dict(seeded_state_dict(model.state_dict())); model = model.cuda()
g = torch.Generator().manual_seed(5)
q, pos, neg, oth = (torch.rand(1, k, 4096, 3, generator=g) * 2 - 1 for k in (1, 2, 14, 1))
nn_dict = {(0, 1): torch.randint(0, 4096, (1024, 1), generator=g).numpy(), (0, 2): torch.randint(0, 4096, (1024, 1), generator=g).numpy()}
opt = torch.optim.Adam(model.parameters(), lr=1e-5)
for _ in range(3):
    training_step(model, opt, q, pos, neg, oth, nn_dict=nn_dict)
rec = []
o_nn, o_kk = train_ops.tgemm_nn, train_ops.tgemm_kk
def t_nn(batch, M, N, K, *a, **k):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); o_nn(batch, M, N, K, *a, **k); e.record()
    rec.append((f"nn kc={int(a[3])} bmode={k.get('bmode', 0)}{' stats' if k.get('stats') is not None else ''}", batch, M, N, K, s, e))
def t_kk(batch, M, N, K, *a, **k):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); o_kk(batch, M, N, K, *a, **k); e.record()
    rec.append((f"kk amode={k.get('amode', 0)} bmode={k.get('bmode', 0)}", batch, M, N, K, s, e))
train_ops.tgemm_nn, train_ops.tgemm_kk = t_nn, t_kk
training_step(model, opt, q, pos, neg, oth, nn_dict=nn_dict)
torch.cuda.synchronize()
agg = collections.OrderedDict()
for kind, b, M, N, K, s, e in rec:
    key = (kind, b, M, N, K)
    a = agg.setdefault(key, [0, 0.0]); a[0] += 1; a[1] += s.elapsed_time(e)
# the first contraction of the backward pass is bracketed by an event recorded before the autograd engine's thread has started feeding the stream:
# its pair measures host time (tens of ms), not the kernel -- listed apart, not in the total
host = {k: v for k, v in agg.items() if v[1] / v[0] > 20.0}
for k in host:
    del agg[k]
tot = sum(v[1] for v in agg.values())
print(f"{len(rec)} GEMM launches, {tot:.2f} ms" + (f" (+ {len(host)} whose event pair timed the host: {[k[:5] for k in host]})" if host else ""))
print(f"{'kind':28s} {'batch':>5s} {'M':>6s} {'N':>6s} {'K':>7s} {'n':>3s} {'us/call':>8s} {'TFLOP/s':>8s} {'GB/s':>7s}")
for (kind, b, M, N, K), (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    us = ms / n * 1e3
    fl = 2.0 * b * M * N * K
    by = 4.0 * b * (M * N + N * K + (M * K if kind.startswith('kk') else 0)) + 4.0 * M * K
    print(f"{kind:28s} {b:5d} {M:6d} {N:6d} {K:7d} {n:3d} {us:8.1f} {fl / us / 1e6:8.1f} {by / us / 1e3:7.0f}")
