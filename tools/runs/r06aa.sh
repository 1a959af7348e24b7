#!/bin/bash
# r06aa: the node's first evaluation on a stream of its own (atlas_rt_eval_event_record) + the operands of Add / Sub in the same pass — parity, then A/B
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
timeout 1800 python -m pytest tests/test_gpu_graph_golden.py tests/test_gpu_graph.py tests/test_gpu_nodes.py tests/test_gpu_one_element.py tests/test_gpu_graph_fuzz.py tests/test_gpu_hardening.py tests/test_gpu_lane_stress.py -q -x -p no:cacheprovider 2>&1 | tail -3 > $O/r06aa_subset.txt
cat $O/r06aa_subset.txt
for v in "" "ATLAS_NO_SIDE_EVAL=1" "" "ATLAS_NO_SIDE_EVAL=1" "" "ATLAS_NO_SIDE_EVAL=1"; do
  env $v timeout 300 python tools/time_graph.py node_einsum,node_relu,node_mul,nanogpt_model,gpt2 2 3 2>&1 | python -c "import sys,json
for l in sys.stdin:
    try: d=json.loads(l); print('[$v]', d['graph'], 'iop', round(d['iop_ms'],2), 'total', round(d['total_ms'],2), d['state'])
    except Exception: pass"
done > $O/r06aa_side_eval_ab.txt 2>&1
cat $O/r06aa_side_eval_ab.txt
