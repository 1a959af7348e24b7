#!/bin/bash
# round 6: enqueue-during-waits variant of the 2^22 step + the bind pass's wait split on the device's clock
O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_sumcheck.py tests/test_gpu_full_size.py -q -x -p no:cacheprovider -k "sumcheck" 2>&1 | tail -3 > $O/r06g_subset.txt
cat $O/r06g_subset.txt
for v in "" "ATLAS_SC_ENQUEUE_ALL=1" "" "ATLAS_SC_ENQUEUE_ALL=1" "" "ATLAS_SC_ENQUEUE_ALL=1"; do
  env $v timeout 300 python bench.py --no-pmc --no-node --no-graph --no-msm --no-cpu-baseline --steps 40 2>/dev/null | python -c "import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('[$v] ms_per_step', round(d['ms_per_step'],4), 'frac', round(r['frac'],4), 'pass_ms', round(r['pass_ms'],4), 'fs_ms', round(r['fs_ms'],4))"
done > $O/r06g_step_ab.txt 2>&1
cat $O/r06g_step_ab.txt
ATLAS_TRACE_CH=1 timeout 120 python bench.py --no-pmc --no-node --no-graph --no-msm --no-cpu-baseline --steps 3 --warmup 1 2>&1 | grep -A24 "channel dot n=22" | tail -26 > $O/r06g_trace_ch.txt
head -14 $O/r06g_trace_ch.txt
ATLAS_DEV_STAMPS=1 timeout 200 python tools/bind_wait_split.py > $O/r06g_bind_wait_split.txt 2>&1
ATLAS_SC_ENQUEUE_ALL=1 ATLAS_DEV_STAMPS=1 timeout 200 python tools/bind_wait_split.py > $O/r06g_bind_wait_split_enqueue_all.txt 2>&1
cat $O/r06g_bind_wait_split.txt; cat $O/r06g_bind_wait_split_enqueue_all.txt
