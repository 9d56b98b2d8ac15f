"""Launch time of the finest FP level's chain: exact fp32 MFMA kernel vs the opt-in split-fp16 ("f32x3") kernel; B = 32, HIP events, 50 launches, best of 5."""
import sys
import torch
sys.path.insert(0, ".")
from patchaugnet_amd.engine import _Chain
from tests.test_gpu_chain import make_layers

B, n, m, c2, c1 = 32, 4096, 1024, 256, 3
_, eng = make_layers([c2 + c1, 256, 256, 256], seed=3)
g = torch.Generator().manual_seed(1)
known = torch.randn(B, m, c2, generator=g).cuda()
skip = torch.randn(B, n, c1, generator=g).cuda()
base = torch.randint(0, m, (B, n // 4, 1), generator=g).repeat_interleave(4, 1)
idx3 = ((base + torch.randint(0, 8, (B, n, 3), generator=g)) % m).int().cuda()
w3 = torch.rand(B, n, 3, generator=g)
w3 = (w3 / w3.sum(-1, keepdim=True)).cuda().contiguous()
gk = torch.randn(B * m, 256, device="cuda")
for name, x3 in (("fp32 MFMA (chain_kernel<1,16,FPX>)", False), ("split fp16 operands (fpx3_kernel)", True)):
    ch = _Chain(eng)
    ch.build_premul(c2, c1, x3=x3)
    best = 1e9
    for rep in range(5):
        ch.fp_premul(known, idx3, w3, skip, B, n, m, c2, c1, g_pre=gk)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            ch.fp_premul(known, idx3, w3, skip, B, n, m, c2, c1, g_pre=gk)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 50 * 1e3)
    print(f"{name}: {best:.1f} us per launch", flush=True)
