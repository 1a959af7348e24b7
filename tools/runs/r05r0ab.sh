cd jolt-atlas_amd; cp libatlas_hip.so libatlas_hip_new.so; cd ..
for rep in 1 2 3; do
  for which in prev new; do
    cp jolt-atlas_amd/libatlas_hip_$which.so jolt-atlas_amd/libatlas_hip.so
    python bench.py --no-pmc --no-node --no-graph --no-msm --no-cpu-baseline --steps 30 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$which step', round(d['ms_per_step'],4), 'frac', round(d['roofline']['frac'],3), 'pass_ms', round(d['roofline']['pass_ms'],4), 'fs_ms', round(d['roofline']['fs_ms'],4))"
  done
done
cp jolt-atlas_amd/libatlas_hip_new.so jolt-atlas_amd/libatlas_hip.so
