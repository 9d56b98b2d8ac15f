"""Which pa_tgemm_nn calls of the module path (eval, 8192-point clouds) differ between the LDS-resident-weights kernel and the LDS-tiled one?
python tools/probes/cm_bisect.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from patchaugnet_amd import _lib, configs, patch_aug_net, train_ops as T
from patchaugnet_amd.weights import seeded_state_dict, synthetic_submaps
lib = _lib.lib()
lib.pa_tgemm_cm_enable.argtypes, lib.pa_tgemm_cm_enable.restype = [ctypes.c_int], None
npts, sampling = 8192, [2048, 256, 32]
cfg = configs.patch_aug_net_config()
cfg["NUM_POINTS"], cfg["SAMPLING"], cfg["MAX_SAMPLES"] = npts, sampling, [sampling[1], sampling[0], npts]
m = patch_aug_net.Network(param=cfg)
m.load_state_dict(seeded_state_dict(m.state_dict())); m = m.cuda().eval()
x = torch.cat([synthetic_submaps(2, npts, 41, "uniform"), synthetic_submaps(1, npts, 42, "street")]).cuda()
orig = T.tgemm_nn
def both(batch, M, N, K, A, sAb, lda, kc, B, sBb, ldb, C, sCb, ldc, **kw):
    C0 = C.clone()
    lib.pa_tgemm_cm_enable(0)
    orig(batch, M, N, K, A, sAb, lda, kc, B, sBb, ldb, C0, sCb, ldc, **kw)
    lib.pa_tgemm_cm_enable(-1)
    orig(batch, M, N, K, A, sAb, lda, kc, B, sBb, ldb, C, sCb, ldc, **kw)
    torch.cuda.synchronize()
    d = (C - C0).abs()
    err = d.max().item()
    flag = "  <<<<<<" if err > 1e-4 * max(C0.abs().max().item(), 1.0) else ""
    desc = {k: (v is not None if not isinstance(v, (int, float)) else v) for k, v in kw.items()}
    print(f"batch={batch} M={M} N={N} K={K} kc={int(kc)} lda={lda} sBb={sBb} ldb={ldb} sCb={sCb} ldc={ldc} {desc}: max diff {err:.3e}{flag}")
    if flag:
        bad = (d > 1e-4).nonzero()
        print("   first bad", bad[:5].tolist(), "count", len(bad), "of", d.numel(), "cols bad (mod 64):", sorted(set((bad[:, -1] % 64).tolist()))[:20], "rows:", sorted(set(bad[:, -2].tolist()))[:20])
T.tgemm_nn = both
with torch.no_grad():
    torch.manual_seed(3)
    m(x, use_engine=False)
