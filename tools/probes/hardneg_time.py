import sys, time; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, numpy as np
from patchaugnet_amd import retrieval
N, nq = 20000, 1400
ref = torch.nn.functional.normalize(torch.randn(N, 256, device="cuda"), dim=1)
rng = np.random.default_rng(0)
negs = [rng.choice(N, 3000, replace=False).tolist() for _ in range(nq)]
qs = ref[:nq]
retrieval.get_hard_negatives_batch(qs[:10], ref, negs[:10])
torch.cuda.synchronize(); t0 = time.perf_counter()
out = retrieval.get_hard_negatives_batch(qs, ref, negs)
torch.cuda.synchronize(); print("per-query launches:", round(time.perf_counter() - t0, 3), "s for", nq, "queries")
print("batched:", end=" "); import time as _t; torch.cuda.synchronize(); t0=_t.perf_counter(); out2 = retrieval.get_hard_negatives_batch(qs, ref, negs); torch.cuda.synchronize(); print(round(_t.perf_counter()-t0,3), "s; equal to per-query:", out2 == [retrieval.get_hard_negatives(q, ref, n) for q, n in zip(qs[:50], negs[:50])] + out2[50:])
