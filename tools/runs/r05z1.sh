mkdir -p gpurun_out; O=$PWD/gpurun_out; R=$PWD; export TMPDIR=/tmp
( cd /tmp && rm -rf /tmp/prof_g2 && ATLAS_GRAPH_VERIFY=0 timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_g2 -o r -- python $R/tools/time_graph.py gpt2 2 2 > /tmp/prof_g2.log 2>&1 )
DB=$(find /tmp/prof_g2 -name "*.db" | head -1)
python tools/rocprof_summary.py $DB $O/r05z1_gpt2_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python tools/time_graph.py gpt2 2 2 (two proofs)" > /dev/null
head -40 $O/r05z1_gpt2_kernel_stats.csv | cut -c1-170
