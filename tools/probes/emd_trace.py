"""Per-round kernel durations and gaps of one EMD forward (16, 4096, 3), 1024 rounds, from a rocprofv3 kernel trace:
   rocprofv3 --kernel-trace -d OUT -o emd -- python tools/probes/emd_trace.py run ; python tools/probes/emd_trace.py OUT/..results.db"""
import os, sys
sys.path.insert(0, os.getcwd())
if sys.argv[1] == "run":
    import torch
    from patchaugnet_amd import emd_module
    g = torch.Generator().manual_seed(11)
    p1, p2 = (torch.rand(16, 4096, 3, generator=g).cuda() for _ in range(2))
    f = emd_module.emdModule()
    for _ in range(2):
        f(p1, p2, 0.02, 1024)
    torch.cuda.synchronize()
else:
    import sqlite3
    c = sqlite3.connect(sys.argv[1])
    view = [n for n, in c.execute("select name from sqlite_master where type='view' and name like 'kernels%'")][-1]
    rows = [r for r in c.execute(f"select name, start, end from {view} order by start") if "emd_round" in r[0]]
    rows = rows[-1024:]
    dur = [(e - s) / 1e3 for _, s, e in rows]
    gap = [(rows[i + 1][1] - rows[i][2]) / 1e3 for i in range(len(rows) - 1)]
    for lo, hi in ((0, 16), (16, 64), (64, 256), (256, 512), (512, 1024)):
        d, g_ = dur[lo:hi], gap[lo:min(hi, len(gap))]
        print(f"rounds {lo:4d}-{hi:4d}: kernel avg {sum(d) / len(d):6.2f} us (min {min(d):5.2f} max {max(d):6.2f}), gap to next avg {sum(g_) / len(g_):5.2f} us")
    print(f"total kernels {sum(dur) / 1e3:.2f} ms, gaps {sum(gap) / 1e3:.2f} ms")
