#!/usr/bin/env python3
"""Per-dispatch durations and algorithmic bytes of the dot-product sumcheck data passes (rocprofv3 --kernel-trace,
rocpd .db) -> CSV.  The passes of one 2^n instance come in a fixed order: k_dot_eval2_f9 over 2^n coefficients
(reads 2 * 2^n * 32 B), then k_dot_bind_eval2_f9 at len = 2^n, 2^(n-1), ... (reads 2 * len * 32 B, writes len * 32 B).
usage: rocprof_dispatch_csv.py <results.db> <n_vars> <out.csv> [note...]"""
import csv
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
start = "start" if "start" in cols else [c for c in cols if "start" in c][0]
end = "end" if "end" in cols else [c for c in cols if "end" in c][0]
rows = list(db.execute(f"select {name_col}, {start}, {end} from kernels where {name_col} like '%k_dot_%' order by {start}"))
out = []
length = None
for name, s, e in rows:
    short = name.split("(")[0]
    if "k_dot_eval" in short:
        length = 1 << n
        nbytes = 2 * length * 32
    elif "k_dot_bind_eval" in short and length:
        nbytes = 2 * length * 32 + length * 32
        length //= 2
    else:
        nbytes = 0
    us = (e - s) / 1e3
    out.append((short, f"{us:.2f}", nbytes, f"{nbytes / us / 1e6:.3f}" if nbytes and us > 0 else ""))
with open(sys.argv[3], "w", newline="") as f:
    if len(sys.argv) > 4:
        f.write("# " + " ".join(sys.argv[4:]) + "\n")
    w = csv.writer(f)
    w.writerow(["kernel", "duration_us", "algorithmic_bytes", "TB_per_s"])
    w.writerows(out)
print(f"{len(out)} dispatches -> {sys.argv[3]}")
