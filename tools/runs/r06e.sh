#!/bin/bash
# round 6: raw column sums from tail_reduce — byte-exactness on the instance / node / model tests, then timing
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ra.py tests/test_gpu_psshout.py tests/test_gpu_graph_golden.py tests/test_gpu_nodes.py tests/test_gpu_batched.py -q -p no:cacheprovider 2>&1 | tail -5 > $O/r06e_subset.txt
cat $O/r06e_subset.txt
for rep in 1 2 3; do
  timeout 300 python tools/time_graph.py node_einsum,node_relu,node_mul,nanogpt_model,gpt2 2 3 2>&1 | python -c "import sys,json
for l in sys.stdin:
    try: d=json.loads(l); print(d['graph'], round(d['iop_ms'],2), round(d['total_ms'],1), d['state'])
    except Exception: pass"
done > $O/r06e_time.txt 2>&1
cat $O/r06e_time.txt
ATLAS_DEV_STAMPS=1 timeout 300 python tools/dev_stamps.py node_einsum > $O/r06e_stamps_einsum.txt 2>&1; tail -3 $O/r06e_stamps_einsum.txt
