#!/bin/bash
# r06p: the capture at HEAD — the whole GPU suite, bench line, rocprofv3 kernel stats of the same command, component / node / MSM / HyperKZG timings (capture_round.sh),
# the device-idle table of a nanoGPT-shaped proof, the reduction's stage trace
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; O=$PWD/gpurun_out; R=$PWD; mkdir -p $O
bash tools/capture_round.sh r06p
( cd /tmp && timeout 300 rocprofv3 --kernel-trace -d /tmp/prof_n -o r -- python $R/tools/time_graph.py nanogpt_model 2 2 > /tmp/prof_n.log 2>&1 )
DB=$(find /tmp/prof_n -name "*.db" | head -1)
python tools/rocprof_gaps.py $DB 480 60 > $O/r06p_nanogpt_gaps.txt 2>&1
ATLAS_TRACE=1 timeout 300 python tools/time_graph.py nanogpt_model,gpt2 2 1 2>&1 | grep -E "prove_reduced_openings|commit_witness|batched_prove \(|onehot pool|^\{" | cut -c1-420 > $O/r06p_reduction_trace.txt
for rep in 1 2 3; do timeout 300 python tools/time_graph.py node_einsum,node_relu,node_mul,nanogpt_model,gpt2 2 3; done 2>/dev/null | grep "^{" > $O/r06p_graphs.jsonl
python -c "
import json
for l in open('$O/r06p_graphs.jsonl'):
    d=json.loads(l); print(d['graph'], 'trace', round(d['trace_ms'],1), 'commit', round(d['commit_ms'],1), 'iop', round(d['iop_ms'],2), 'reduction', round(d['reduction_ms'],1), 'hkzg', round(d['hyperkzg_ms'],1), 'total', round(d['total_ms'],2), d['state'])
" > $O/r06p_graphs.txt
cat $O/r06p_graphs.txt; head -14 $O/r06p_nanogpt_gaps.txt
