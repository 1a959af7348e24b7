#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; rocpd .db).
FETCH_SIZE on gfx950 reports half the bytes of wide coalesced reads (MI355X_MICROARCH.md §HBM;
confirmed here on __amd_rocclr_copyBuffer: 64 MB reported for a 128 MB copy), so reads are
doubled; WRITE_SIZE is exact on the same calibration.
usage: pmc_summary.py <fetch.db> <write.db> <out.csv> [note...]"""
import csv
import sqlite3
import sys
from collections import defaultdict


def per_kernel(db, counter):
    out = defaultdict(lambda: [0, 0.0])
    for name, val in sqlite3.connect(db).execute(
            "select kernel_name, value from counters_collection where counter_name = ?", (counter,)):
        out[name][0] += 1
        out[name][1] += val
    return out


f = per_kernel(sys.argv[1], "FETCH_SIZE")
w = per_kernel(sys.argv[2], "WRITE_SIZE")
rows = []
for k in sorted(set(f) | set(w), key=lambda k: -(f[k][1] * 2 + w[k][1])):
    calls = max(f[k][0], w[k][0])
    rd = f[k][1] * 2 * 1024          # KB -> bytes, x2 correction
    wr = w[k][1] * 1024
    rows.append((k, calls, rd, wr, (rd + wr) / max(calls, 1)))
with open(sys.argv[3], "w", newline="") as fh:
    if len(sys.argv) > 4:
        fh.write("# " + " ".join(sys.argv[4:]) + "\n")
    cw = csv.writer(fh)
    cw.writerow(["kernel", "launches", "read_bytes(FETCH_SIZE*2)", "write_bytes", "hbm_bytes_per_launch"])
    for r in rows:
        cw.writerow([r[0], r[1], int(r[2]), int(r[3]), int(r[4])])
for r in rows[:12]:
    print("%-70s launches %4d  read %8.1f MB  write %8.1f MB  per-launch %8.2f MB" % (r[0][:70], r[1], r[2] / 1e6, r[3] / 1e6, r[4] / 1e6))
