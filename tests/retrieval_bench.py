#!/usr/bin/env python
"""Retrieval at an Oxford-sized set (BASELINE configs[2] shape): 23 trips x ~130 submaps, 256-D, all trip pairs, top-25 + 1 %.
Times the GPU kNN part and the whole get_recall_precision call, next to the oracle (KD-tree, CPU)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from oracle import recall_cpu
from oracle.gen_recall_golden import synthetic_route
from patchaugnet_amd import retrieval

sizes = [130] * 23
_, desc, tuples = synthetic_route(1, sizes)
d = torch.from_numpy(desc).cuda()
retrieval.get_recall_precision(d, sizes, tuples)          # warm-up
torch.cuda.synchronize()
t0 = time.perf_counter(); res = retrieval.get_recall_precision(d, sizes, tuples); torch.cuda.synchronize(); t_gpu = time.perf_counter() - t0
t0 = time.perf_counter(); ref = recall_cpu.get_recall_precision(desc, sizes, tuples); t_cpu = time.perf_counter() - t0
a, b = retrieval.average(res), recall_cpu.average(ref)
db, q = d[:130].contiguous(), d.contiguous()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20):
    retrieval.hip_knn(d, q, 26)
torch.cuda.synchronize(); t_knn = (time.perf_counter() - t0) / 20
print(f"{len(desc)} descriptors, {len(res)} trip pairs: product {t_gpu*1e3:.0f} ms (host bookkeeping included), oracle KD-tree {t_cpu*1e3:.0f} ms")
print(f"recall@1 product {a[0][0]:.4f} oracle {b[0][0]:.4f} | recall@1% {a[2]:.4f} vs {b[2]:.4f} | max |d recall@N| {np.abs(a[0]-b[0]).max():.2e}")
print(f"brute-force kNN, all {len(desc)} queries vs all {len(desc)} descriptors, k=26: {t_knn*1e3:.2f} ms")
