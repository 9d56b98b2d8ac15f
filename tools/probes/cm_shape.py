"""One pa_tgemm_nn shape, LDS-resident-weights kernel against the LDS-tiled one: where do they differ?  python tools/probes/cm_shape.py B M N K bmode"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from patchaugnet_amd import _lib, train_ops as T
lib = _lib.lib()
lib.pa_tgemm_cm_enable.argtypes, lib.pa_tgemm_cm_enable.restype = [ctypes.c_int], None
B, M, N, K, bmode = (int(a) for a in sys.argv[1:6])
g = torch.Generator().manual_seed(1)
A = (torch.randn(M, K, generator=g) / K ** 0.5).cuda()
X = torch.randn(B, K, N, generator=g).cuda()
aux = torch.randn(B, K, N, generator=g).cuda() if bmode >= 2 else None
p = (torch.rand(7, K, generator=g) + 0.25).cuda() if bmode else None
outs = []
for on in (0, 1):
    lib.pa_tgemm_cm_enable(on)
    C = torch.zeros(B, M, N, device="cuda")
    for _ in range(int(os.environ.get("REPS", "1"))):
        T.tgemm_nn(B, M, N, K, A, 0, K, True, X, K * N, N, C, M * N, N, bmode=bmode, baux=aux, bp=p)
    torch.cuda.synchronize()
    outs.append(C)
d = (outs[0] - outs[1]).abs()
bad = (d > 1e-4 * max(outs[0].abs().max().item(), 1.0)).nonzero()
print(f"B={B} M={M} N={N} K={K} bmode={bmode}: max diff {d.max().item():.3e}, bad {len(bad)} of {d.numel()}")
if len(bad):
    import collections
    tw = int(os.environ.get("TW", "64"))
    tiles = collections.Counter(((b * N + n) // tw) for b, m, n in bad.tolist())
    print(" bad column tiles (global index: count):", sorted(tiles.items())[:40])
    print(" rows:", sorted(set(bad[:, 1].tolist()))[:40])
    print(" cols mod tw:", sorted(set((bad[:, 2] % tw).tolist()))[:40])
