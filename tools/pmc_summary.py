#!/usr/bin/env python
"""Per-kernel average of every PMC counter in a rocprofv3 rocpd database:  python tools/pmc_summary.py x_results.db"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
cols = [d[1] for d in c.execute("pragma table_info(counters_collection)")]
print("columns:", cols)
q = ("select kernel_name, counter_name, count(*), avg(value), min(value), max(value) from counters_collection "
     "group by kernel_name, counter_name order by avg(value) desc")
try:
    for r in c.execute(q):
        print(f"{r[0][:90]:90s} {r[1]:14s} n={r[2]:3d} avg={r[3]:.1f} min={r[4]:.1f} max={r[5]:.1f}")
except Exception as e:
    print("query failed:", e)
    for r in c.execute("select * from counters_collection limit 5"):
        print(r)
