#!/usr/bin/env python
"""Per-dispatch view of a rocprofv3 rocpd database: kernels grouped by (name, grid) with average duration, and the launch sequence of
one step with the gaps between consecutive dispatches.

    python tools/rocprof_trace.py x_results.db [--schema] [--step-marker fps_reg_kernel<256, 16]
"""
import sqlite3
import sys
import collections


def main():
    db = sys.argv[1]
    c = sqlite3.connect(db)
    if "--schema" in sys.argv:
        for name, typ in c.execute("select name, type from sqlite_master where type in ('table','view') order by name"):
            cols = [r[1] for r in c.execute(f"pragma table_info('{name}')")]
            print(typ, name, cols)
        return
    view = None
    for name, in c.execute("select name from sqlite_master where type='view' and name like 'kernels%'"):
        view = name
    cols = [r[1] for r in c.execute(f"pragma table_info('{view}')")]
    gx = "grid_x" if "grid_x" in cols else "grid_size_x"
    gy = "grid_y" if "grid_y" in cols else "grid_size_y"
    wx = "workgroup_x" if "workgroup_x" in cols else "workgroup_size_x"
    rows = list(c.execute(f"select name, {gx}, {gy}, {wx}, start, end from {view} order by start"))
    agg = collections.OrderedDict()
    for name, x, y, w, s, e in rows:
        key = (name[:90], x // max(w, 1), y)
        a = agg.setdefault(key, [0, 0.0])
        a[0] += 1
        a[1] += (e - s) / 1e3
    print("== per (kernel, grid): calls, avg us")
    for (name, x, y), (n, tot) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        if n >= 5:
            print(f"{tot / n:9.2f} us  x{n:4d}  grid=({x},{y})  {name}")
    # launch sequence of the last complete step: from the last-but-one occurrence of the marker kernel to the last one
    marker = "fps_reg_kernel<256, 16"
    if "--step-marker" in sys.argv:
        marker = sys.argv[sys.argv.index("--step-marker") + 1]
    idx = [i for i, r in enumerate(rows) if marker in r[0]]
    if len(idx) >= 3:
        a, b = idx[-3], idx[-2]
        print("== one step, in launch order: start offset us, duration us, gap before us, kernel")
        t0 = rows[a][4]
        prev_end = None
        for name, x, y, w, s, e in rows[a:b]:
            gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
            print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.2f} {gap:7.2f}  grid=({x // max(w, 1)},{y}) {name[:100]}")
            prev_end = e
        print(f"step span {(rows[b][4] - t0) / 1e3:.1f} us")


if __name__ == "__main__":
    main()
