#!/bin/bash
# round 6: where the resident bind-pass launch should start (ATLAS_SC_RESIDENT_LOG), against a launch per round
O=gpurun_out; mkdir -p $O
REPS=4 SIZES=12,13,14,16,18,20,22 timeout 300 python tools/dbg_resident.py | tail -1
for rep in 1 2 3; do
for v in "ATLAS_SC_NO_RESIDENT=1" "ATLAS_SC_RESIDENT_LOG=22" "ATLAS_SC_RESIDENT_LOG=21" "ATLAS_SC_RESIDENT_LOG=20" "ATLAS_SC_RESIDENT_LOG=19" "ATLAS_SC_RESIDENT_LOG=18" "ATLAS_SC_RESIDENT_LOG=17"; do
  env $v timeout 300 python bench.py --no-pmc --no-node --no-graph --no-msm --no-cpu-baseline --steps 40 2>/dev/null | python -c "import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('[$v] ms_per_step', round(d['ms_per_step'],4), 'frac', round(r['frac'],4), 'pass_ms', round(r['pass_ms'],4), 'fs_ms', round(r['fs_ms'],4), 'launches', r.get('launches'))"
done; done > $O/r06i_resident_sweep.txt 2>&1
cat $O/r06i_resident_sweep.txt
