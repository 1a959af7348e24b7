mkdir -p gpurun_out; O=gpurun_out
B="python bench.py --no-msm --no-node --no-graph --no-cpu-baseline --no-pmc --steps 30"
for i in 1 2 3; do
 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('db2 r0=256', d['ms_per_step'], d['roofline']['frac'], d.get('proof_sha16'))"
 ATLAS_F9_R0_BLOCKS=512 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('db2 r0=512', d['ms_per_step'], d['roofline']['frac'])"
 ATLAS_F9_R0_BLOCKS=1024 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('db2 r0=1024', d['ms_per_step'], d['roofline']['frac'])"
done > $O/r05b_round0_ab.txt 2>&1
cat $O/r05b_round0_ab.txt
python -m pytest tests/test_gpu_sumcheck.py tests/test_gpu_full_size.py -q -m gpu -x 2>&1 | tail -3
ATLAS_PROF=1 ATLAS_GRAPH_VERIFY=0 python tools/time_graph.py gpt2 2 2 > $O/r05b_prof_gpt2.txt 2>&1
tail -1 $O/r05b_prof_gpt2.txt | cut -c1-300
