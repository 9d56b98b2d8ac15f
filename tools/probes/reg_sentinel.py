"""Register sentinels (tools/probes/reg_sentinel.hip) on one stream while pa_linear_f16 / pa_linear loop on another: are a resident wave's VGPRs / SGPRs intact?
python tools/probes/reg_sentinel.py [trials]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from patchaugnet_amd import engine
from patchaugnet_amd._lib import call, ptr
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "reg_sentinel.so"))
lib.reg_sentinel_launch.argtypes = [ctypes.c_int, ctypes.c_long, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 10
g = torch.Generator().manual_seed(0)
rows, k, n = 131072, 256, 256
x = torch.randn(rows, k, generator=g).cuda(); wt = (torch.randn(k, n, generator=g) / k ** 0.5).cuda().contiguous(); bias = torch.zeros(n, device="cuda"); out = torch.empty(rows, n, device="cuda")
wp, wp16 = engine.pack_weights(wt), engine.pack_weights_f16(wt)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
for name in ("pa_linear_f16", "pa_linear", None):
    for blocks in (32, 512):
        bv = torch.zeros(1, dtype=torch.int32, device="cuda"); bs = torch.zeros(1, dtype=torch.int32, device="cuda"); ex = torch.zeros(4, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        for t in range(trials):
            lib.reg_sentinel_launch(blocks, 600, bv.data_ptr(), bs.data_ptr(), ex.data_ptr(), sb.cuda_stream)
            if name:
                with torch.cuda.stream(sa):
                    for _ in range(8):
                        call(name, rows, k, n, ptr(x), k, ptr(wt), ptr(wp16 if name.endswith("f16") else wp), ptr(bias), 1, None, 0, ptr(out), n)
            torch.cuda.synchronize()
        print(f"co-runner {str(name):14s} sentinel workgroups {blocks:4d}: changed VGPR values {int(bv)}, changed SGPR values {int(bs)}", ex.tolist() if int(bv) or int(bs) else "")
