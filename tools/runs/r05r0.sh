timeout 1400 python -m pytest tests/test_gpu_sumcheck.py tests/test_gpu_full_size.py tests/test_gpu_nodes.py tests/test_gpu_graph_golden.py tests/test_gpu_einsum.py -q -m gpu -x 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_r0 -o r -- python $R/bench.py --no-pmc --no-node --no-graph --no-msm --no-cpu-baseline --steps 10 > /tmp/prof_r0.log 2>&1
DB=$(find /tmp/prof_r0 -name "*.db" | head -1)
cd $R && python tools/rocprof_dispatch_csv.py $DB 22 gpurun_out/r05r0_dispatches.csv "per-dispatch data passes of the 2^22 instances, round 0 with one reduction per four products" > /dev/null 2>&1
python tools/rocprof_summary.py $DB gpurun_out/r05r0_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python bench.py --no-pmc --no-node --no-graph --no-msm --steps 10" > /dev/null 2>&1
grep "k_dot_eval2_f9" gpurun_out/r05r0_kernel_stats.csv | cut -c1-200; head -14 gpurun_out/r05r0_dispatches.csv | cut -c1-200
