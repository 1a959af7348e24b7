#!/bin/bash
# r06v: the reduction's host side — batched inversions under the fold, one visit per shared key, lock-free job hand-off to the host threads
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_reduced_openings.py tests/test_gpu_opening.py tests/test_gpu_rlc.py tests/test_gpu_graph_golden.py tests/test_gpu_batched.py -q -x -p no:cacheprovider 2>&1 | tail -3
for ht in 8 16 32; do for spin in 300 2000; do
  echo "== ATLAS_HOST_THREADS=$ht ATLAS_HOST_SPIN_US=$spin"
  ATLAS_HOST_THREADS=$ht ATLAS_HOST_SPIN_US=$spin ATLAS_TRACE=1 timeout 300 python tools/time_graph.py gpt2 2 2 2>&1 | grep -a "batched_prove (8\|prove_reduced_openings batched\|prove_reduced_openings inst\|onehot pool\|total_ms" | cut -c1-420 | tail -5 | sed -e 's/"n_nodes.*//'
done; done > $O/r06v_reduction_threads.txt 2>&1
cat $O/r06v_reduction_threads.txt
