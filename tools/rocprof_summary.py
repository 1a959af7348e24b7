#!/usr/bin/env python3
"""Dump the per-kernel summary (rocprofv3 --kernel-trace --stats, rocpd .db output) as CSV.
usage: rocprof_summary.py <results.db> <out.csv> [note...]"""
import csv
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
with open(sys.argv[2], "w", newline="") as f:
    if len(sys.argv) > 3:
        f.write("# " + " ".join(sys.argv[3:]) + "\n")
    w = csv.writer(f)
    w.writerow(["kernel", "calls", "total_us", "avg_us", "pct"])
    for r in rows:
        w.writerow([r[0], r[1], f"{r[2]:.3f}", f"{r[3]:.3f}", f"{r[4]:.2f}"])
print(open(sys.argv[2]).read())
