#!/bin/bash
# round 6: (1) byte-exactness + timing of the two-table clamp phases and the prebuilt PS instances; (2) device stamps of a node; (3) xdist trial
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_psshout.py tests/test_gpu_graph_golden.py tests/test_gpu_nodes.py tests/test_gpu_one_element.py -q -x -p no:cacheprovider 2>&1 | tail -5 > $O/r06d_subset.txt
cat $O/r06d_subset.txt
for v in "" "ATLAS_PS_NO_DUP=1" "ATLAS_NO_PREBUILD=1" "ATLAS_PS_NO_DUP=1 ATLAS_NO_PREBUILD=1"; do
  for rep in 1 2; do
  env $v timeout 300 python tools/time_graph.py node_einsum,node_relu,node_mul,nanogpt_model,gpt2 2 3 2>&1 | python -c "import sys,json
for l in sys.stdin:
    try: d=json.loads(l); print('[$v]', d['graph'], round(d['iop_ms'],2), round(d['total_ms'],1), d['state'])
    except Exception: pass"
  done
done > $O/r06d_ab.txt 2>&1
cat $O/r06d_ab.txt
ATLAS_DEV_STAMPS=1 timeout 300 python tools/dev_stamps.py node_einsum > $O/r06d_stamps_einsum.txt 2>&1; tail -3 $O/r06d_stamps_einsum.txt
