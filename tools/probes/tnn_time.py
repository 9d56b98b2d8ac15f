import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from patchaugnet_amd._lib import call, ptr
B, n, m = 32, 4096, 1024
u = (torch.rand(B, n, 3, device="cuda") * 2 - 1); k = u[:, :m].contiguous()
w = torch.empty(B, n, 3, device="cuda"); idx = torch.empty(B, n, 3, dtype=torch.int32, device="cuda")
for _ in range(5): call("pa_three_nn_weights", B, n, m, ptr(u), ptr(k), ptr(w), ptr(idx))
torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): call("pa_three_nn_weights", B, n, m, ptr(u), ptr(k), ptr(w), ptr(idx))
e1.record(); torch.cuda.synchronize()
print(os.environ.get("PA_3NN_VARIANT", "0"), f"{e0.elapsed_time(e1)/50*1e3:.1f} us", int(idx.sum()))
