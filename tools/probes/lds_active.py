"""tools/probes/lds_active.hip (the selection plumbing of the sampling kernel, self-checking) on one stream while pa_linear_f16 / pa_linear loop on another.
python tools/probes/lds_active.py [trials]"""
import ctypes, os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from patchaugnet_amd import engine
from patchaugnet_amd._lib import call, ptr
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "lds_active.so"))
lib.lds_active_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 4
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 10
g = torch.Generator().manual_seed(0)
rows, k, n = 131072, 256, 256
x = torch.randn(rows, k, generator=g).cuda(); wt = (torch.randn(k, n, generator=g) / k ** 0.5).cuda().contiguous(); bias = torch.zeros(n, device="cuda"); out = torch.empty(rows, n, device="cuda")
wp, wp16 = engine.pack_weights(wt), engine.pack_weights_f16(wt)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
blocks = 32
for name in ("pa_linear_f16", "pa_linear", None):
    bad = torch.zeros(blocks, dtype=torch.int32, device="cuda"); alloc = torch.zeros(blocks, dtype=torch.int32, device="cuda"); ex = torch.zeros(blocks * 2, dtype=torch.int64, device="cuda")
    seen = collections.Counter(); badalloc = collections.Counter()
    torch.cuda.synchronize()
    for t in range(trials):
        bad.zero_(); torch.cuda.synchronize()
        if name:
            with torch.cuda.stream(sa):
                for _ in range(8):
                    call(name, rows, k, n, ptr(x), k, ptr(wt), ptr(wp16 if name.endswith("f16") else wp), ptr(bias), 1, None, 0, ptr(out), n)
        lib.lds_active_launch(blocks, 1023, 1, bad.data_ptr(), alloc.data_ptr(), ex.data_ptr(), sb.cuda_stream)
        torch.cuda.synchronize()
        for a, b in zip(alloc.tolist(), bad.tolist()):
            seen[a & 0xffffffff] += 1
            if b: badalloc[a & 0xffffffff] += 1
    print(f"co-runner {str(name):14s}: LDS_ALLOC values seen {{{', '.join(f'0x{k:08x}: {v}' for k, v in sorted(seen.items()))}}}")
    print(f"    workgroups with mismatching rounds by LDS_ALLOC {{{', '.join(f'0x{k:08x}: {v}' for k, v in sorted(badalloc.items()))}}}")
    nz = bad.nonzero().flatten().tolist()
    if nz:
        i = nz[0]; print(f"    last trial e.g. workgroup {i}: {int(bad[i])} thread-rounds wrong, got 0x{int(ex[2*i]) & (2**64-1):016x} want 0x{int(ex[2*i+1]) & (2**64-1):016x}")
