#!/bin/bash
# r06w: the dense members of a reduction in one pool (DensePool) — parity, then A/B against an instance each
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_reduced_openings.py tests/test_gpu_opening.py tests/test_gpu_graph_golden.py tests/test_gpu_batched.py tests/test_gpu_graph.py tests/test_gpu_one_element.py -q -x -p no:cacheprovider 2>&1 | tail -3
for rep in 1 2; do for v in "" "ATLAS_NO_DENSE_POOL=1"; do
  echo "== [$v]"
  env $v ATLAS_HOST_THREADS=16 ATLAS_TRACE=1 timeout 300 python tools/time_graph.py nanogpt_model,gpt2 2 2 2>&1 | grep -a "batched_prove (\|prove_reduced_openings batched\|prove_reduced_openings inst\|total_ms" | cut -c1-420 | tail -12 | sed -e 's/"n_nodes.*//'
done; done > $O/r06w_dense_pool_ab.txt 2>&1
cat $O/r06w_dense_pool_ab.txt
