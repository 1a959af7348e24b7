#!/bin/bash
# round 6: the resident bind-pass kernel — parity, then the step
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_sumcheck.py tests/test_gpu_full_size.py tests/test_gpu_sharded.py tests/test_gpu_fuzz.py tests/test_gpu_einsum_layouts.py -q -x -p no:cacheprovider 2>&1 | tail -6 > $O/r06h_subset.txt
cat $O/r06h_subset.txt
for v in "" "ATLAS_SC_NO_RESIDENT=1" "" "ATLAS_SC_NO_RESIDENT=1" "" "ATLAS_SC_NO_RESIDENT=1"; do
  env $v timeout 300 python bench.py --no-pmc --no-node --no-graph --no-msm --no-cpu-baseline --steps 40 2>/dev/null | python -c "import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('[$v] ms_per_step', round(d['ms_per_step'],4), 'frac', round(r['frac'],4), 'pass_ms', round(r['pass_ms'],4), 'fs_ms', round(r['fs_ms'],4), 'launches', r.get('launches'))"
done > $O/r06h_step_ab.txt 2>&1
cat $O/r06h_step_ab.txt
ATLAS_TRACE_CH=1 timeout 120 python bench.py --no-pmc --no-node --no-graph --no-msm --no-cpu-baseline --steps 3 --warmup 1 2>&1 | grep -A24 "channel dot n=22" | tail -26 > $O/r06h_trace_ch.txt
head -24 $O/r06h_trace_ch.txt
