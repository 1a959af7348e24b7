#!/bin/bash
# round 6: the transcript's compression on AVX2 — KATs through the suite's transcript / sumcheck tests, then A/B on whole proofs
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_sumcheck.py tests/test_gpu_graph_golden.py tests/test_gpu_batched.py tests/test_gpu_hyperkzg.py -q -x -p no:cacheprovider 2>&1 | tail -3 > $O/r06m_subset.txt
cat $O/r06m_subset.txt
for rep in 1 2 3; do
for v in "" "ATLAS_BLAKE_PORTABLE=1"; do
  env $v timeout 300 python tools/time_graph.py node_einsum,node_relu,nanogpt_model,gpt2 2 3 2>&1 | python -c "import sys,json
for l in sys.stdin:
    try: d=json.loads(l); print('[$v]', d['graph'], round(d['iop_ms'],2), round(d['total_ms'],1), d['state'])
    except Exception: pass"
done; done > $O/r06m_ab.txt 2>&1
cat $O/r06m_ab.txt
ATLAS_PROF=1 timeout 300 python tools/time_graph.py gpt2 2 2 > $O/r06m_gpt2_host_prof.txt 2>&1
