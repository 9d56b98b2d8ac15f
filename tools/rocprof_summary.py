#!/usr/bin/env python
"""Turn a rocprofv3 rocpd database (--kernel-trace --stats) into the per-kernel summary CSV kept under profiles/.

    python tools/rocprof_summary.py gpurun_out/prof/x_results.db profiles/rNN_name_kernel_stats.csv
"""
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


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
