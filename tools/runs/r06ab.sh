#!/bin/bash
# r06ab: counter blocks from a zeroed pool, one memset in the sign scan — parity (incl. the abort / timeout tests and the leak test), then A/B
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
timeout 1800 python -m pytest tests/test_gpu_psshout.py tests/test_gpu_ra.py tests/test_gpu_hardening.py tests/test_gpu_leaks.py tests/test_gpu_lifecycle.py tests/test_gpu_lane_stress.py tests/test_gpu_graph_golden.py tests/test_gpu_softmax.py tests/test_gpu_nodes.py -q -x -p no:cacheprovider 2>&1 | tail -3 > $O/r06ab_subset.txt
cat $O/r06ab_subset.txt
for v in "" "ATLAS_NO_COUNTER_POOL=1" "" "ATLAS_NO_COUNTER_POOL=1" "" "ATLAS_NO_COUNTER_POOL=1"; do
  env $v timeout 300 python tools/time_graph.py node_einsum,node_relu,nanogpt_model,gpt2 2 3 2>&1 | python -c "import sys,json
for l in sys.stdin:
    try: d=json.loads(l); print('[$v]', d['graph'], 'iop', round(d['iop_ms'],2), 'total', round(d['total_ms'],2), d['state'])
    except Exception: pass"
done > $O/r06ab_counter_pool_ab.txt 2>&1
cat $O/r06ab_counter_pool_ab.txt
