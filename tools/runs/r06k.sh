#!/bin/bash
# round 6: Booleanity's expanding table in one launch + nothing enqueued inside the host-only prefix of a one-hot batch: parity, then A/B
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ra.py tests/test_gpu_batched.py tests/test_gpu_graph_golden.py tests/test_gpu_nodes.py tests/test_gpu_one_element.py -q -x -p no:cacheprovider 2>&1 | tail -3 > $O/r06k_subset.txt
cat $O/r06k_subset.txt
for rep in 1 2 3; do
for v in "" "ATLAS_LOOKAHEAD_FLAT=1 ATLAS_BOOL_EXPAND_PER_ROUND=1"; do
  env $v timeout 300 python tools/time_graph.py node_einsum,node_relu,nanogpt_model,gpt2 2 3 2>&1 | python -c "import sys,json
for l in sys.stdin:
    try: d=json.loads(l); print('[$v]', d['graph'], round(d['iop_ms'],2), round(d['total_ms'],1), d['state'])
    except Exception: pass"
done; done > $O/r06k_ab.txt 2>&1
cat $O/r06k_ab.txt
