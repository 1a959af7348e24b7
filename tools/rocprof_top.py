#!/usr/bin/env python3
"""Top kernels of a rocprofv3 rocpd .db with short names.  usage: rocprof_top.py <results.db> [rows]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
for name, calls, total, avg, pct in db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"):
    short = name.replace("(anonymous namespace)::", "").replace("atlas::", "")
    short = re.sub(r"^void ", "", short)
    short = re.split(r"\((?![^<]*>)", short)[0][:60]
    print(f"{short:60s} {calls:6d} {total / 1e3:10.2f} ms total {avg:10.1f} us avg {pct:6.2f}%")
    n -= 1
    if n <= 0:
        break
