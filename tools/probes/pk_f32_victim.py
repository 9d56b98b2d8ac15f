"""tools/probes/pk_f32_victim.hip on one stream while pa_linear_f16 / pa_linear loop on another: do packed and scalar fp32 arithmetic agree in every lane and round?
python tools/probes/pk_f32_victim.py [trials]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from patchaugnet_amd import engine
from patchaugnet_amd._lib import call, ptr
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "pk_f32_victim.so"))
lib.pk_victim_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 10
g = torch.Generator().manual_seed(0)
rows, k, n = 131072, 256, 256
x = torch.randn(rows, k, generator=g).cuda(); wt = (torch.randn(k, n, generator=g) / k ** 0.5).cuda().contiguous(); bias = torch.zeros(n, device="cuda"); out = torch.empty(rows, n, device="cuda")
wp, wp16 = engine.pack_weights(wt), engine.pack_weights_f16(wt)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
for name in ("pa_linear_f16", "pa_linear", None):
    for blocks, lds in ((32, 65600), (256, 65600), (1024, 1024)):
        bad = torch.zeros(1, dtype=torch.int32, device="cuda"); ex = torch.zeros(8, device="cuda")
        torch.cuda.synchronize()
        for t in range(trials):
            if name:
                with torch.cuda.stream(sa):
                    for _ in range(8):
                        call(name, rows, k, n, ptr(x), k, ptr(wt), ptr(wp16 if name.endswith("f16") else wp), ptr(bias), 1, None, 0, ptr(out), n)
            lib.pk_victim_launch(int(os.environ.get('VICTIM_FORM', '0')), blocks, 1500, lds, bad.data_ptr(), ex.data_ptr(), sb.cuda_stream)
            torch.cuda.synchronize()
        print(f"co-runner {str(name):14s} victim workgroups {blocks:4d} x {lds} B LDS: packed != scalar in {int(bad)} of {trials * blocks * 256 * 1500 * 8} lane-rounds",
              [f"{v:.9g}" for v in ex.tolist()] if int(bad) else "")
