mkdir -p gpurun_out; O=$PWD/gpurun_out; R=$PWD; export TMPDIR=/tmp
for blocks in 0 512 1024; do
( cd /tmp && rm -rf /tmp/prof_q$blocks && ATLAS_F9_R0_BLOCKS=$blocks timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_q$blocks -o r -- python $R/bench.py --no-pmc --no-node --no-graph --no-msm --no-cpu-baseline --steps 10 > /tmp/prof_q$blocks.log 2>&1 )
DB=$(find /tmp/prof_q$blocks -name "*.db" | head -1)
python tools/rocprof_dispatch_csv.py $DB 22 $O/r05q_dispatches_r0_$blocks.csv "round-0 grid cap $blocks (0 = 256)" > /dev/null
grep "k_dot_eval2_f9" $O/r05q_dispatches_r0_$blocks.csv | awk -F, '{s+=$2; n++} END {print "r0 cap '$blocks': k_dot_eval2_f9 avg us", s/n, "n", n}'
tail -1 /tmp/prof_q$blocks.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['frac'])"
done
