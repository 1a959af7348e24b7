#!/bin/bash
# r06u: the reduction's kernels after r06t — parity of the reduction / golden graphs, stage trace, per-kernel device time of one GPT-2-shaped proof
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_reduced_openings.py tests/test_gpu_opening.py tests/test_gpu_rlc.py tests/test_gpu_graph_golden.py -q -x -p no:cacheprovider 2>&1 | tail -3
ATLAS_TRACE=1 timeout 300 python tools/time_graph.py gpt2 2 2 2>&1 | grep -a "prove_reduced_openings\|batched_prove (8\|onehot pool\|total_ms" | cut -c1-420 | tail -7 > $O/r06u_reduction_trace.txt
cat $O/r06u_reduction_trace.txt
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_u -o r -- python $OLDPWD/tools/time_graph.py gpt2 2 1 > /tmp/prof_u.log 2>&1 )
DB=$(find /tmp/prof_u -name "*.db" | head -1)
python tools/rocprof_summary.py $DB $O/r06u_gpt2_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python tools/time_graph.py gpt2 2 1 (two proofs: divide calls and totals by 2)" > /dev/null
grep -h "k_pool\|k_rlc\|k_g1_sum" $O/r06u_gpt2_kernel_stats.csv | awk -F'",' '{n=split($1,a,"("); print a[1] a[2] "  |  " $2}' | cut -c1-140
grep -h "k_pool_eq_full\|k_pool_gather" $O/r06u_gpt2_kernel_stats.csv | cut -c1-40,80-140
