#!/usr/bin/env python
"""Retrieval kNN on MFMA at database scale: 20 000 x 20 000 x 256-D, k = 26 (knn_cuda.knn_mfma_raw), wall time per call."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from patchaugnet_amd import knn_cuda
n, dim, k = 20000, 256, 26
g = torch.Generator(device="cuda").manual_seed(1)
ref = torch.randn(dim, n, device="cuda", generator=g); q = torch.randn(dim, n, device="cuda", generator=g)
for _ in range(2): knn_cuda.knn_mfma_raw(ref, q, k)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): knn_cuda.knn_mfma_raw(ref, q, k)
torch.cuda.synchronize(); print(f"knn_mfma {n} x {n} x {dim}, k = {k}: {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms per call")
