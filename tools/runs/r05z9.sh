# one 2^22 table MSM under the kernel trace: the timeline of the last call
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
LOG_N=22 TAB_C=0 timeout 400 rocprofv3 --kernel-trace -d /tmp/prof_msm -o msm -- python $R/tools/time_msm_tab.py 2>&1 | tail -4
DB=$(find /tmp/prof_msm -name "*.db" | head -1)
python $R/tools/rocprof_timeline.py $DB 30 > $R/gpurun_out/r05z9_msm_timeline.txt 2>&1
cat $R/gpurun_out/r05z9_msm_timeline.txt
cd $R && python tools/time_graph.py node_relu,node_einsum,node_mul,microgpt_model 2 4 2>&1 | tail -4 | cut -c1-220
