#!/bin/bash
# r06y: reduction after the host-side work (one visit per shared key, batched inversions, dense pool, sparse address rounds, host tables in parallel)
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_reduced_openings.py tests/test_gpu_opening.py tests/test_gpu_graph_golden.py tests/test_gpu_batched.py tests/test_gpu_graph.py tests/test_gpu_one_element.py tests/test_gpu_sharded.py tests/test_gpu_fuzz.py -q -x -p no:cacheprovider 2>&1 | tail -3 > $O/r06y_subset.txt
cat $O/r06y_subset.txt
for rep in 1 2 3; do
  timeout 300 python tools/time_graph.py node_einsum,node_relu,nanogpt_model,gpt2 2 3 2>&1 | python -c "import sys,json
for l in sys.stdin:
    try: d=json.loads(l); print(d['graph'], 'commit', round(d['commit_ms'],1), 'iop', round(d['iop_ms'],2), 'reduction', round(d['reduction_ms'],1), 'hkzg', round(d['hyperkzg_ms'],1), 'total', round(d['total_ms'],2), d['state'])
    except Exception: pass"
done > $O/r06y_graphs.txt 2>&1
cat $O/r06y_graphs.txt
ATLAS_TRACE=1 timeout 300 python tools/time_graph.py gpt2 2 2 2>&1 | grep -a "batched_prove (8\|prove_reduced_openings\|onehot pool" | cut -c1-640 | tail -6 > $O/r06y_reduction_trace.txt
cat $O/r06y_reduction_trace.txt
