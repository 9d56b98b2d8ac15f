#!/usr/bin/env python
"""Turn a rocprofv3 rocpd database (--kernel-trace --stats) into the per-kernel summary CSVs kept under profiles/.

    python tools/rocprof_summary.py gpurun_out/prof/x_results.db profiles/rNN_name_kernel_stats.csv

Writes the overall per-kernel table (rocprofv3's own top_kernels view) and, next to it, `*_by_grid.csv`: the same kernels split by launch grid.
bench.py runs more than one batch size in one process (the headline's batch 32 and the reference protocol's batch 100, datasets/scene_dataset.py:
666-686), so a kernel's OVERALL average mixes launches of different sizes; the roofline's launch is the row with the batch-32 grid.
"""
import collections
import csv
import sqlite3
import sys


def main(db, out):
    c = sqlite3.connect(db)
    rows = list(c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_us", "avg_us", "percent"])
        for name, calls, tot, avg, pct in rows:
            w.writerow([name, calls, f"{tot:.3f}", f"{avg:.3f}", f"{pct:.3f}"])
    print(f"{len(rows)} kernels -> {out}")
    try:
        view = None
        for name, in c.execute("select name from sqlite_master where type='view' and name like 'kernels%'"):
            view = name
        cols = [r[1] for r in c.execute(f"pragma table_info('{view}')")]
        gx = "grid_x" if "grid_x" in cols else "grid_size_x"
        gy = "grid_y" if "grid_y" in cols else "grid_size_y"
        wx = "workgroup_x" if "workgroup_x" in cols else "workgroup_size_x"
        agg = collections.OrderedDict()
        for name, x, y, wg, s, e in c.execute(f"select name, {gx}, {gy}, {wx}, start, end from {view}"):
            a = agg.setdefault((name, x // max(wg, 1), y), [0, 0.0, 1e30, 0.0])
            d = (e - s) / 1e3
            a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
        out2 = out[:-4] + "_by_grid.csv" if out.endswith(".csv") else out + "_by_grid.csv"
        with open(out2, "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["kernel", "workgroups_x", "grid_y", "calls", "total_us", "avg_us", "min_us", "max_us"])
            for (name, x, y), (n, tot, mn, mx) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
                w.writerow([name, x, y, n, f"{tot:.3f}", f"{tot / n:.3f}", f"{mn:.3f}", f"{mx:.3f}"])
        print(f"{len(agg)} (kernel, grid) rows -> {out2}")
    except Exception as ex:      # the per-dispatch view is a convenience: never lose the main table over it
        print("by-grid table skipped:", repr(ex))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
