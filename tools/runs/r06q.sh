#!/bin/bash
# r06q: per-kernel split of the 2^22 fixed-base MSM (rocprofv3 --kernel-trace --stats on tools/time_msm_tab.py)
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
( cd /tmp && LOG_N=22 TAB_C=0 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_q -o r -- python $OLDPWD/tools/time_msm_tab.py > /tmp/prof_q.log 2>&1 )
DB=$(find /tmp/prof_q -name "*.db" | head -1)
python tools/rocprof_summary.py $DB $O/r06q_msm22_kernel_stats.csv "LOG_N=22 TAB_C=0 rocprofv3 --kernel-trace --stats -- python tools/time_msm_tab.py" > /dev/null
head -40 $O/r06q_msm22_kernel_stats.csv; tail -5 /tmp/prof_q.log
