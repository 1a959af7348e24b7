#!/bin/bash
# round 6: lazy enqueue of the 2^22 passes (review item 3) + join without lane waits: parity, A/B of the step, dispatch trace
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_sumcheck.py tests/test_gpu_full_size.py tests/test_gpu_sharded.py tests/test_gpu_ra.py tests/test_gpu_batched.py -q -x -p no:cacheprovider 2>&1 | tail -4 > $O/r06f_subset.txt
cat $O/r06f_subset.txt
for v in "" "ATLAS_SC_ENQUEUE_ALL=1" "" "ATLAS_SC_ENQUEUE_ALL=1"; do
  env $v timeout 300 python bench.py --no-pmc --no-node --no-graph --no-msm --no-cpu-baseline --steps 40 2>/dev/null | python -c "import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('[$v] ms_per_step', round(d['ms_per_step'],4), 'frac', round(r['frac'],4), 'pass_ms', round(r['pass_ms'],4), 'fs_ms', round(r['fs_ms'],4))"
done > $O/r06f_step_ab.txt 2>&1
cat $O/r06f_step_ab.txt
ATLAS_TRACE_CH=1 timeout 120 python bench.py --no-pmc --no-node --no-graph --no-msm --no-cpu-baseline --steps 3 --warmup 1 2>&1 | grep -A24 "channel dot n=22" | tail -26 > $O/r06f_trace_ch.txt
ATLAS_SC_ENQUEUE_ALL=1 ATLAS_TRACE_CH=1 timeout 120 python bench.py --no-pmc --no-node --no-graph --no-msm --no-cpu-baseline --steps 3 --warmup 1 2>&1 | grep -A24 "channel dot n=22" | tail -26 > $O/r06f_trace_ch_all.txt
head -8 $O/r06f_trace_ch.txt; head -8 $O/r06f_trace_ch_all.txt
for v in "" "ATLAS_JOIN_WAIT=1" "" "ATLAS_JOIN_WAIT=1"; do
  env $v timeout 300 python tools/time_graph.py node_einsum,nanogpt_model,gpt2 2 3 2>&1 | python -c "import sys,json
for l in sys.stdin:
    try: d=json.loads(l); print('[$v]', d['graph'], round(d['iop_ms'],2), round(d['total_ms'],1), d['state'])
    except Exception: pass"
done > $O/r06f_join_ab.txt 2>&1
cat $O/r06f_join_ab.txt
