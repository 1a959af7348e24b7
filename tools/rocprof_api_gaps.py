#!/usr/bin/env python3
"""What the HOST does while the device idles: for the largest (kernel before -> kernel after) idle pairs of a
`rocprofv3 --kernel-trace --hip-trace` run (rocpd .db; no --pmc in that run), the HIP API calls whose start falls inside the gap,
summed per API name over all occurrences of the pair.
usage: rocprof_api_gaps.py <results.db> [span_ms] [top_pairs]"""
import collections
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
span = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 1e12
top = int(sys.argv[3]) if len(sys.argv) > 3 else 8
tables = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]


def cols_of(t):
    return [r[1] for r in db.execute(f"pragma table_info({t})")]


kc = cols_of("kernels")
kname = "name" if "name" in kc else [c for c in kc if "name" in c][0]
rows = list(db.execute(f"select {kname}, start, end from kernels order by start"))
t_end = rows[-1][2]
rows = [r for r in rows if r[1] >= t_end - span]
# the API regions: a view named regions / regions_and_samples / api with (name, start, end); print what exists when none matches
api = None
for t in ("regions", "regions_and_samples", "api", "rocpd_region"):
    if t in tables and {"start", "end"} <= set(cols_of(t)):
        api = t
        break
if api is None:
    print("no API region table; tables:", tables)
    for t in tables:
        print(t, cols_of(t))
    sys.exit(0)
ac = cols_of(api)
aname = "name" if "name" in ac else [c for c in ac if "name" in c][0]
calls = sorted(db.execute(f"select {aname}, start, end from {api}"), key=lambda r: r[1])
starts = [c[1] for c in calls]


def short(name):
    s = re.sub(r"^void ", "", name.replace("(anonymous namespace)::", "").replace("atlas::", ""))
    return re.split(r"\((?![^<]*>)", s)[0][:36]


import bisect
gaps = collections.defaultdict(list)
hi, hi_name = rows[0][1], None
for name, s, e in rows:
    if s > hi and hi_name is not None:
        gaps[(hi_name, short(name))].append((hi, s))
    if e > hi:
        hi, hi_name = e, short(name)
print(f"API table `{api}`: {len(calls)} calls; {len(rows)} dispatches")
for (a, b), occ in sorted(gaps.items(), key=lambda kv: -sum(y - x for x, y in kv[1]))[:top]:
    tot = sum(y - x for x, y in occ) / 1e3
    print(f"\n{tot / 1e3:9.3f} ms  x{len(occ):<5d} {tot / len(occ):8.1f} us each   {a}  ->  {b}")
    per = collections.defaultdict(lambda: [0, 0.0])
    for x, y in occ:
        i = bisect.bisect_left(starts, x)
        # calls that started before the gap and are still running at its start (a wait the host is in) count too
        j = i - 1
        while j >= 0 and i - j < 50:
            if calls[j][2] > x:
                per["(running at gap start) " + calls[j][0]][0] += 1
                per["(running at gap start) " + calls[j][0]][1] += (min(calls[j][2], y) - x) / 1e3
            j -= 1
        while i < len(calls) and calls[i][1] < y:
            per[calls[i][0]][0] += 1
            per[calls[i][0]][1] += (min(calls[i][2], y) - calls[i][1]) / 1e3
            i += 1
    for n, (c, us) in sorted(per.items(), key=lambda kv: -kv[1][1])[:10]:
        print(f"      {us / len(occ):8.1f} us per gap  ({c / len(occ):5.1f} calls)  {n}")
