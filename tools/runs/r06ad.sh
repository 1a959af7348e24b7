#!/bin/bash
# r06ad: the pool's first cycle round through the index rows (no materialised H = F[idx]) — parity, the reduction's trace, memory
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_reduced_openings.py tests/test_gpu_opening.py tests/test_gpu_graph_golden.py tests/test_gpu_graph.py tests/test_gpu_sharded.py tests/test_gpu_one_element.py tests/test_gpu_fuzz.py -q -x -p no:cacheprovider 2>&1 | tail -3 > $O/r06ad_subset.txt
cat $O/r06ad_subset.txt
for rep in 1 2 3; do timeout 300 python tools/time_graph.py nanogpt_model,gpt2 2 3 2>&1 | python -c "import sys,json
for l in sys.stdin:
    try: d=json.loads(l); print(d['graph'], 'reduction', round(d['reduction_ms'],1), 'total', round(d['total_ms'],2), d['state'])
    except Exception: pass"; done > $O/r06ad_graphs.txt
cat $O/r06ad_graphs.txt
ATLAS_TRACE=1 timeout 300 python tools/time_graph.py gpt2 2 2 2>&1 | grep -a "batched_prove (8\|prove_reduced_openings\|onehot pool" | cut -c1-300 | tail -6 > $O/r06ad_reduction_trace.txt
cat $O/r06ad_reduction_trace.txt
{ for w in 1 4; do echo "== gpt2 world $w"; timeout 400 python tools/time_sharded.py gpt2 $w 2 2>&1 | grep "RANK" | cut -c1-300; done; } > $O/r06ad_sharded_memory.txt 2>&1
cat $O/r06ad_sharded_memory.txt
