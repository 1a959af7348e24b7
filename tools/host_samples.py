#!/usr/bin/env python3
"""Profile of the host thread from an ATLAS_HOST_SAMPLE file (csrc/host_sampler.hpp): frames `library+0xoffset`, leaf first.
Offsets inside libatlas_hip.so are resolved against the symbol table of the SAME build (llvm-nm); other libraries stay as library names.
Prints the exclusive profile (leaf frame), the inclusive profile of libatlas_hip.so functions, and — for the samples whose leaf is outside
the library (libc memcpy / malloc, the HIP runtime) — the nearest libatlas_hip.so caller.
usage: host_samples.py <samples.txt> [path/to/libatlas_hip.so] [top]"""
import bisect
import collections
import os
import subprocess
import sys

path = sys.argv[1]
so = sys.argv[2] if len(sys.argv) > 2 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "jolt-atlas_amd", "libatlas_hip.so")
top = int(sys.argv[3]) if len(sys.argv) > 3 else 45
nm = subprocess.run(["nm", "-C", "--defined-only", "-n", so], capture_output=True, text=True).stdout
syms = []
for l in nm.split("\n"):
    p = l.split(" ", 2)
    if len(p) == 3 and p[1] in "tTwW":
        syms.append((int(p[0], 16), p[2]))
syms.sort()
addrs = [a for a, _ in syms]


def resolve(frame):
    lib, _, off = frame.partition("+")
    if "libatlas_hip" not in lib:
        return lib
    i = bisect.bisect_right(addrs, int(off, 16)) - 1
    name = syms[i][1] if i >= 0 else frame
    return name.split("(")[0][-90:] if not name.startswith("(anonymous") else name.replace("(anonymous namespace)::", "").split("(")[0][-90:]


excl, incl, via = collections.Counter(), collections.Counter(), collections.Counter()
n = 0
for line in open(path):
    if line.startswith("#") or not line.strip():
        continue
    fr = [resolve(f) for f in line.split()]
    n += 1
    excl[fr[0]] += 1
    seen = set()
    for f in fr:
        if f not in seen and not f.endswith(".so") and ".so." not in f and f != "python3.10" and not f.startswith("?"):
            seen.add(f); incl[f] += 1
    if fr[0].endswith(".so") or ".so." in fr[0]:
        caller = next((f for f in fr[1:] if not (f.endswith(".so") or ".so." in f or f.startswith("?") or f == "python3.10")), "?")
        via[(fr[0], caller)] += 1
print(f"{n} samples ({n * 0.05:.1f} ms at 50 us)")
print("\n-- exclusive (leaf)")
for k, v in excl.most_common(top):
    print(f"{100.0 * v / n:6.2f} %  {v * 0.05:8.2f} ms  {k}")
print("\n-- inclusive, functions of libatlas_hip.so")
for k, v in incl.most_common(top):
    print(f"{100.0 * v / n:6.2f} %  {v * 0.05:8.2f} ms  {k}")
print("\n-- leaf outside the library: (library, nearest caller inside)")
for (l, c), v in via.most_common(top):
    print(f"{100.0 * v / n:6.2f} %  {v * 0.05:8.2f} ms  {l:28s} <- {c}")
