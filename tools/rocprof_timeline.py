#!/usr/bin/env python3
"""Timeline of the last N kernel dispatches of a rocprofv3 rocpd .db: start offset, duration, gap to the previous
dispatch (us).  usage: rocprof_timeline.py <results.db> [N]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 80
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
start = "start" if "start" in cols else [c for c in cols if "start" in c][0]
end = "end" if "end" in cols else [c for c in cols if "end" in c][0]
rows = list(db.execute(f"select {name_col}, {start}, {end} from kernels order by {start}"))[-n:]
t0, prev = rows[0][1], None
for name, s, e in rows:
    short = re.sub(r"^void ", "", name.replace("(anonymous namespace)::", "").replace("atlas::", ""))
    short = re.split(r"\((?![^<]*>)", short)[0][:44]
    gap = (s - prev) / 1e3 if prev else 0.0
    print(f"{(s - t0) / 1e3:10.1f} us  {short:44s} {(e - s) / 1e3:8.1f} us   gap {gap:7.1f}")
    prev = e
