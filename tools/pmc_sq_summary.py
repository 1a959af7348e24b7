#!/usr/bin/env python3
"""SQ counters per kernel from one `rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY
SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE` run (rocpd .db): per kernel NAME (substring filters on the command line) the number
of dispatches, the summed counters and the fractions of SQ_WAVE_CYCLES — valu = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES (the convention of
profiles/r02f_*_pmc_sq.txt: 0.70 for the largest pass at one wavefront per SIMD), wait_any = SQ_WAIT_ANY / SQ_WAVE_CYCLES, wait_inst =
SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES.
usage: pmc_sq_summary.py <results.db> <out.txt> <name substring> [...]      (PMC runs never carry --stats / other trace domains)"""
import collections
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
want = sys.argv[3:]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
n_disp = collections.Counter()
for name, counter, val in db.execute("select kernel_name, counter_name, value from counters_collection"):
    if want and not any(w in name for w in want):
        continue
    m = re.search(r"k_\w+(<[^>(]*>)?", name)                      # the kernel's own name (names in an anonymous namespace start with a parenthesis)
    short = (m.group(0) if m else name.split("(")[0].replace("void ", "").replace("atlas::", ""))[:48]
    acc[short][counter] += val
    if counter == "SQ_WAVES":
        n_disp[short] += 1
lines = ["# " + " ".join(sys.argv[1:2]), "%-48s %6s %14s %8s %8s %8s %8s %14s" % ("kernel", "disp", "wave_cycles", "valu", "any", "wait_any", "wait_ins", "valu_insts")]
for k, c in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    wc = c.get("SQ_WAVE_CYCLES", 0) or 1.0
    lines.append("%-48s %6d %14.0f %8.3f %8.3f %8.3f %8.3f %14.0f" % (k, n_disp[k], wc, c.get("SQ_ACTIVE_INST_VALU", 0) / wc, c.get("SQ_ACTIVE_INST_ANY", 0) / wc,
                                                                   c.get("SQ_WAIT_ANY", 0) / wc, c.get("SQ_WAIT_INST_ANY", 0) / wc, c.get("SQ_INSTS_VALU", 0)))
open(sys.argv[2], "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:14]))
