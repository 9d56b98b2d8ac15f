"""Do two kernels of different streams share the chip?  The finest FP chain (MFMA-bound, 135 KB of LDS per CU) on one stream and a geometry kernel on
another: time of the pair issued together against the sum of the two alone.  usage: python tools/probes/corun.py"""
import sys
import torch
sys.path.insert(0, ".")
from patchaugnet_amd import _lib, pointops
from patchaugnet_amd._lib import call, ptr
from patchaugnet_amd.engine import _Chain
from patchaugnet_amd.weights import synthetic_submaps
from tests.test_gpu_chain import make_layers

B, n, m, c2, c1 = 32, 4096, 1024, 256, 3
_, eng = make_layers([c2 + c1, 256, 256, 256], seed=3)
g = torch.Generator().manual_seed(1)
known = torch.randn(B, m, c2, generator=g).cuda()
skip = torch.randn(B, n, c1, generator=g).cuda()
idx3 = torch.randint(0, m, (B, n, 3), generator=g).int().cuda()
w3 = torch.rand(B, n, 3, generator=g)
w3 = (w3 / w3.sum(-1, keepdim=True)).cuda().contiguous()
gk = torch.randn(B * m, 256, device="cuda")
ch = _Chain(eng)
ch.build_premul(c2, c1)
x = synthetic_submaps(B, n, 5, "uniform").squeeze(1).cuda().contiguous()
q = x[:, :m].contiguous()
kidx = torch.empty(B, m, 20, dtype=torch.int32, device="cuda"); kd = torch.empty(B, m, 20, device="cuda")
tidx = torch.empty(B, n, 3, dtype=torch.int32, device="cuda"); tw = torch.empty(B, n, 3, device="cuda")

def chain():
    ch.fp_premul(known, idx3, w3, skip, B, n, m, c2, c1, g_pre=gk)

def knn():
    call("pa_knnquery", B, n, m, 20, ptr(x), ptr(q), ptr(kidx), ptr(kd))

def tnn():
    call("pa_three_nn_weights", B, n, m, ptr(x), ptr(q), ptr(tw), ptr(tidx))

s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

def timed(fa, fb, reps=20):
    best = 1e9
    for _ in range(5):
        torch.cuda.synchronize()
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record()
        s1.wait_event(e0); s2.wait_event(e0)
        with torch.cuda.stream(s1):
            for _ in range(reps):
                if fa: fa()
            e1.record()
        with torch.cuda.stream(s2):
            for _ in range(reps):
                if fb: fb()
            e2.record()
        torch.cuda.synchronize()
        best = min(best, max(e0.elapsed_time(e1), e0.elapsed_time(e2)) / reps * 1e3)
    return best

for f in (chain, knn, tnn):
    f()
torch.cuda.synchronize()
tc, tk, tt = timed(chain, None), timed(None, knn), timed(None, tnn)
print(f"alone: fp0 chain {tc:.1f} us, kNN (sa0 shape) {tk:.1f} us, 3-NN (fp0 shape) {tt:.1f} us")
print(f"chain || kNN : {timed(chain, knn):.1f} us per pair (sum {tc + tk:.1f})")
print(f"chain || 3-NN: {timed(chain, tnn):.1f} us per pair (sum {tc + tt:.1f})")
print(f"kNN   || 3-NN: {timed(knn, tnn):.1f} us per pair (sum {tk + tt:.1f})")
