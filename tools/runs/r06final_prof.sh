#!/bin/bash
# r06final_prof: rocprofv3 --kernel-trace --stats of the bench command at HEAD (the dominant kernel's average duration next to r06final_bench.json's roofline)
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; O=$PWD/gpurun_out; mkdir -p $O
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_f -o r -- python $OLDPWD/bench.py --no-pmc --no-node --steps 10 > /tmp/prof_f.log 2>&1 )
DB=$(find /tmp/prof_f -name "*.db" | head -1)
python tools/rocprof_summary.py $DB $O/r06final_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python bench.py --no-pmc --no-node --steps 10" > /dev/null
python tools/rocprof_dispatch_csv.py $DB 22 $O/r06final_dispatches.csv "per-dispatch data passes of the 2^22 instances of the same run"
grep "k_dot_" $O/r06final_kernel_stats.csv | cut -c1-60,200-300 | head -6; head -4 $O/r06final_dispatches.csv | cut -c1-200
