#!/usr/bin/env python
"""Per-kernel averages of the PMC counters in a rocprofv3 rocpd sqlite db."""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute("pragma table_info('counters_collection')")]
q = ("select kernel_name, counter_name, avg(value), count(*) from counters_collection group by kernel_name, counter_name"
     if "kernel_name" in cols else None)
if q is None:
    print("columns:", cols)
    sys.exit(0)
for r in c.execute(q):
    print(f"{r[0][:60]:60s} {r[1]:28s} avg={r[2]:.4g} n={r[3]}")
