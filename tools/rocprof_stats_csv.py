#!/usr/bin/env python3
"""Kernel statistics of a `rocprofv3 --kernel-trace --stats` run as CSV (ROCm 7.2 writes a rocpd SQLite database):
python tools/rocprof_stats_csv.py <dir with *_results.db> > profiles/rNN_bench_kernel_stats.csv"""
import glob
import os
import sqlite3
import sys

db = glob.glob(os.path.join(sys.argv[1], "**", "*_results.db"), recursive=True)[0]
c = sqlite3.connect(db)
print("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs,VGPRs,LDS")
rows = c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), max(vgpr_count), max(lds_size) "
                 "from kernels group by name order by sum(duration) desc").fetchall()
total = sum(r[2] for r in rows)
for name, calls, tot, avg, mn, mx, vg, lds in rows:
    print(f'"{name}",{calls},{tot},{avg:.1f},{100.0 * tot / total:.2f},{mn},{mx},{vg},{lds}')
