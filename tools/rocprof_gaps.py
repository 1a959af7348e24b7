#!/usr/bin/env python3
"""Where the device idles: over the last `span_ms` of a rocprofv3 kernel trace (rocpd .db), the idle time between
dispatches (no kernel of any stream running), summed per (kernel before -> kernel after) pair.
usage: rocprof_gaps.py <results.db> [span_ms] [top]"""
import collections
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
span = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 1e12
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
start = "start" if "start" in cols else [c for c in cols if "start" in c][0]
end = "end" if "end" in cols else [c for c in cols if "end" in c][0]
rows = list(db.execute(f"select {name_col}, {start}, {end} from kernels order by {start}"))
t_end = rows[-1][2]
rows = [r for r in rows if r[1] >= t_end - span]


def short(name):
    s = re.sub(r"^void ", "", name.replace("(anonymous namespace)::", "").replace("atlas::", ""))
    return re.split(r"\((?![^<]*>)", s)[0][:36]


gaps = collections.defaultdict(lambda: [0, 0.0])
busy = 0.0
hi, hi_name = rows[0][1], None
for name, s, e in rows:
    if s > hi:
        if hi_name is not None:
            g = gaps[(hi_name, short(name))]
            g[0] += 1; g[1] += (s - hi) / 1e3
        busy += 0
    if e > hi:
        busy += (e - max(hi, s)) / 1e3
        hi, hi_name = e, short(name)
total = (rows[-1][2] - rows[0][1]) / 1e3
idle = sum(v[1] for v in gaps.values())
print(f"window {total / 1e3:.3f} ms, {len(rows)} dispatches, device busy {busy / 1e3:.3f} ms, idle {idle / 1e3:.3f} ms")
for (a, b), (n, us) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"{us / 1e3:9.3f} ms  x{n:<5d} {us / n:8.1f} us each   {a}  ->  {b}")
