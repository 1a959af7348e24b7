#!/bin/bash
# r06t: F9 mixed addition in the one-hot commitments, k_pool_hist without atomics, the binned joint-polynomial kernel — parity subset, then timings
cd "${GRAFT_REPO_ROOT:-.}"; O=gpurun_out; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_reduced_openings.py tests/test_gpu_opening.py tests/test_gpu_rlc.py tests/test_gpu_msm.py tests/test_gpu_hyperkzg.py tests/test_gpu_graph_golden.py tests/test_gpu_graph.py -q -x -p no:cacheprovider 2>&1 | tail -4 > $O/r06t_subset.txt
cat $O/r06t_subset.txt
for rep in 1 2; do
for v in "" "ATLAS_RLC_NO_BINS=1"; do
  env $v timeout 300 python tools/time_graph.py nanogpt_model,gpt2 2 3 2>&1 | python -c "import sys,json
for l in sys.stdin:
    try: d=json.loads(l); print('[$v]', d['graph'], 'commit', round(d['commit_ms'],1), 'iop', round(d['iop_ms'],1), 'reduction', round(d['reduction_ms'],1), 'hkzg', round(d['hyperkzg_ms'],1), 'total', round(d['total_ms'],1), d['state'])
    except Exception: pass"
done; done > $O/r06t_ab.txt 2>&1
cat $O/r06t_ab.txt
ATLAS_TRACE=1 timeout 300 python tools/time_graph.py gpt2 2 2 2>&1 | grep -a "prove_reduced_openings\|batched_prove (8\|onehot pool" | cut -c1-420 | tail -6 > $O/r06t_reduction_trace.txt
cat $O/r06t_reduction_trace.txt
