cd jolt-atlas_amd; cp libatlas_hip.so libatlas_hip_new.so; cd ..
for rep in 1 2 3; do
  for which in prev new; do
    cp jolt-atlas_amd/libatlas_hip_$which.so jolt-atlas_amd/libatlas_hip.so
    ATLAS_GRAPH_VERIFY=0 python tools/time_graph.py gpt2,nanogpt_model 2 3 2>&1 | grep "^{" | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('$which', d['graph'], round(d['total_ms'],1), 'iop', round(d['iop_ms'],1), 'red', round(d['reduction_ms'],1), 'kzg', round(d['hyperkzg_ms'],1), 'commit', round(d['commit_ms'],1))"
  done
done
cp jolt-atlas_amd/libatlas_hip_new.so jolt-atlas_amd/libatlas_hip.so
