#!/bin/bash
# r06s: per-kernel device time of ONE GPT-2-shaped proof (rocprofv3 --kernel-trace --stats on tools/time_graph.py gpt2 2 1 = one warm-up + one timed proof)
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_s -o r -- python $OLDPWD/tools/time_graph.py gpt2 2 1 > /tmp/prof_s.log 2>&1 )
DB=$(find /tmp/prof_s -name "*.db" | head -1)
python tools/rocprof_summary.py $DB $O/r06s_gpt2_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python tools/time_graph.py gpt2 2 1 (two proofs: divide calls and totals by 2)" > /dev/null
head -45 $O/r06s_gpt2_kernel_stats.csv | cut -c1-200; tail -3 /tmp/prof_s.log | cut -c1-300
