#!/bin/bash
# r06ac: the reduction's members freed before the opening, the opening's large arenas released when it ends — per-rank peak device memory of the
# GPT-2-shaped proof at world 1 / 2 / 4 (one GPU), the memory trace, and the cost (A/B ATLAS_KEEP_ARENAS=1)
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
{ for w in 1 2 4; do echo "== gpt2 world $w"; timeout 400 python tools/time_sharded.py gpt2 $w 3 2>&1 | grep "RANK\|Error\|error" | cut -c1-400; done;
  echo "== gpt2 world 4, ATLAS_KEEP_ARENAS=1"; ATLAS_KEEP_ARENAS=1 timeout 400 python tools/time_sharded.py gpt2 4 3 2>&1 | grep "RANK\|Error\|error" | cut -c1-400; } > $O/r06ac_sharded_memory.txt 2>&1
cat $O/r06ac_sharded_memory.txt
ATLAS_MEM_TRACE=1 timeout 300 python tools/time_graph.py gpt2 2 1 2>&1 | grep -a "atlas mem" | tail -10 > $O/r06ac_mem_trace.txt
cat $O/r06ac_mem_trace.txt
for v in "" "ATLAS_KEEP_ARENAS=1" "" "ATLAS_KEEP_ARENAS=1" "" "ATLAS_KEEP_ARENAS=1"; do
  env $v timeout 300 python tools/time_graph.py nanogpt_model,gpt2 2 3 2>&1 | python -c "import sys,json
for l in sys.stdin:
    try: d=json.loads(l); print('[$v]', d['graph'], 'reduction', round(d['reduction_ms'],1), 'hkzg', round(d['hyperkzg_ms'],1), 'total', round(d['total_ms'],2), 'wall', round(d['wall_ms'],1), d['state'])
    except Exception: pass"
done > $O/r06ac_arenas_ab.txt 2>&1
cat $O/r06ac_arenas_ab.txt
timeout 900 python -m pytest tests/test_gpu_reduced_openings.py tests/test_gpu_sharded.py tests/test_gpu_graph_golden.py tests/test_gpu_hyperkzg.py tests/test_gpu_leaks.py tests/test_gpu_full_size.py -q -x -p no:cacheprovider 2>&1 | tail -2
