#!/usr/bin/env python
"""Concurrency picture of a rocprofv3 --kernel-trace run: for the last 60 % of the trace (steady state), wall time, sum of kernel
durations, time with >= 1 / >= 2 kernels running, and per-kernel share.   python tools/trace_overlap.py x_results.db"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
view = "kernels" if "kernels" in tabs else None
if view is None:
    print("tables:", tabs); sys.exit(1)
cols = [d[1] for d in c.execute(f"pragma table_info({view})")]
name_col = "name" if "name" in cols else "kernel_name"
rows = list(c.execute(f"select {name_col}, start, end from {view} order by start"))
import re
rows = [r for r in rows if "at::native" not in r[0] and "rocclr" not in r[0]][-700:]      # the last ~25 steps of our own kernels
ev = []
for n, s, e in rows:
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
busy1 = busy2 = 0; depth = 0; last = ev[0][0]
for t, d in ev:
    if depth >= 1: busy1 += t - last
    if depth >= 2: busy2 += t - last
    depth += d; last = t
wall = ev[-1][0] - ev[0][0]
tot = sum(e - s for _, s, e in rows)
print(f"steady-state window {wall/1e6:.2f} ms: sum of kernel durations {tot/1e6:.2f} ms ({tot/wall:.2f}x), >=1 kernel running {busy1/wall:.1%}, >=2 running {busy2/wall:.1%}")
agg = {}
for n, s, e in rows:
    m = re.search(r"(\w+_kernel(<[^>]*>)?)", n)
    k = m.group(1) if m else n[:50]
    agg[k] = agg.get(k, 0) + (e - s)
for k, v in sorted(agg.items(), key=lambda kv: -kv[1])[:12]:
    print(f"  {v/wall:6.1%} of wall  {k}")
