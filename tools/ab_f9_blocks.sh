# A/B of the grid cap of the big data passes (ATLAS_F9_BIG_BLOCKS / ATLAS_F9_BIG_MIN_LOG, atlas_hip.hip): interleaved repetitions
for rep in 1 2 3 4 5; do
  for cfg in "ATLAS_F9_BIG_BLOCKS=256" "ATLAS_F9_BIG_BLOCKS=512" "ATLAS_F9_BIG_BLOCKS=512 ATLAS_F9_BIG_MIN_LOG=20" "ATLAS_F9_BIG_BLOCKS=384"; do
    env $cfg python bench.py --no-pmc --no-node --no-graph --no-msm --no-cpu-baseline --steps 40 --warmup 5 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); r=d['roofline']
print('%-55s ms_per_step %.4f pass_ms %.4f frac %.3f' % ('$cfg', d['ms_per_step'], r.get('pass_ms',0), r['frac']))"
  done
done
