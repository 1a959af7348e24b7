#!/usr/bin/env python3
"""Per-dispatch durations of the kernels whose name contains <substr> (rocprofv3 rocpd .db).
usage: rocprof_dispatches.py <results.db> <substr> [max_rows]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
start = "start" if "start" in cols else [c for c in cols if "start" in c][0]
end = "end" if "end" in cols else [c for c in cols if "end" in c][0]
lim = int(sys.argv[3]) if len(sys.argv) > 3 else 40
q = f"select {name_col}, {start}, {end} from kernels where {name_col} like ? order by {start} limit {lim}"
for n, s, e in db.execute(q, (f"%{sys.argv[2]}%",)):
    print(f"{n.split('(')[0]:40s} {(e - s) / 1e3:10.1f} us")
