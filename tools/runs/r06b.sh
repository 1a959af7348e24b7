#!/bin/bash
# round 6: A/B of the lane streams at HEAD, and the host profile of the fused-c_attn GPT-2 graph
O=gpurun_out; mkdir -p $O
for rep in 1 2; do
  timeout 300 python tools/time_graph.py node_einsum,nanogpt_model,gpt2 2 3 2>&1 | python -c "import sys,json
for l in sys.stdin:
    try: d=json.loads(l); print('lanes ', d['graph'], round(d['iop_ms'],1), round(d['total_ms'],1))
    except Exception: pass"
  ATLAS_NO_LANE_STREAMS=1 timeout 300 python tools/time_graph.py node_einsum,nanogpt_model,gpt2 2 3 2>&1 | python -c "import sys,json
for l in sys.stdin:
    try: d=json.loads(l); print('1strm ', d['graph'], round(d['iop_ms'],1), round(d['total_ms'],1))
    except Exception: pass"
done > $O/r06b_lane_ab.txt 2>&1
cat $O/r06b_lane_ab.txt
ATLAS_PROF=1 timeout 300 python tools/time_graph.py gpt2 2 2 > $O/r06b_gpt2_host_prof.txt 2>&1
grep -c prof $O/r06b_gpt2_host_prof.txt
