# Same-box A/B of two builds of the library over whole proofs (through gpurun from the repo root).  Beside jolt-atlas_amd/libatlas_hip.so (the new
# build) it expects jolt-atlas_amd/libatlas_hip_prev.so: check out the sources of the commit to compare against, `make -C jolt-atlas_amd`, copy the
# result to that name, restore the sources, `make` again.  (*.so is git-ignored but travels to the GPU box.)
cd jolt-atlas_amd; cp libatlas_hip.so libatlas_hip_new.so; cd ..
for rep in 1 2 3 4; do
  for which in prev new; do
    cp jolt-atlas_amd/libatlas_hip_$which.so jolt-atlas_amd/libatlas_hip.so
    python tools/time_graph.py nanogpt_model,gpt2_layer 3 2 2>&1 | grep "^{" | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('$which', d['graph'], round(d['total_ms'],1), 'iop', round(d['iop_ms'],1))"
  done
done
cp jolt-atlas_amd/libatlas_hip_new.so jolt-atlas_amd/libatlas_hip.so
