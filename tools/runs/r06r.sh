#!/bin/bash
# r06r: the opening reduction of the GPT-2-shaped proof — stage trace (ATLAS_TRACE) and the host-thread count of its batched sumcheck
cd "${GRAFT_REPO_ROOT:-.}"; O=gpurun_out; mkdir -p $O
nproc > $O/r06r_reduction_threads.txt
for ht in 8 16 32; do
  echo "== ATLAS_HOST_THREADS=$ht" >> $O/r06r_reduction_threads.txt
  ATLAS_HOST_THREADS=$ht timeout 300 python tools/time_graph.py gpt2,nanogpt_model 2 3 2>&1 | python -c "import sys,json
for l in sys.stdin:
    try: d=json.loads(l); print(d['graph'], 'iop', round(d['iop_ms'],1), 'reduction', round(d['reduction_ms'],1), 'total', round(d['total_ms'],1), d['state'])
    except Exception: pass" >> $O/r06r_reduction_threads.txt
done
ATLAS_TRACE=1 timeout 300 python tools/time_graph.py gpt2 2 2 2>&1 | grep -a "prove_reduced_openings\|batched_prove (8\|onehot pool" | cut -c1-420 > $O/r06r_reduction_trace.txt
cat $O/r06r_reduction_threads.txt; tail -6 $O/r06r_reduction_trace.txt
