#!/bin/bash
# r06final: the last commit of the round that touches code — smoke(), the whole GPU suite, bench.py, three alternations of the graphs
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; O=$PWD/gpurun_out; mkdir -p $O
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 > $O/r06final_smoke.txt; cat $O/r06final_smoke.txt
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -6 > $O/r06final_pytest_gpu.txt; cat $O/r06final_pytest_gpu.txt
timeout 900 python bench.py > $O/r06final_bench.json 2> $O/r06final_bench.err; tail -c 400 $O/r06final_bench.json
for rep in 1 2 3; do timeout 300 python tools/time_graph.py node_einsum,node_relu,node_mul,nanogpt_model,gpt2 2 3; done 2>/dev/null | grep "^{" | python -c "
import sys, json
for l in sys.stdin:
    d=json.loads(l); print(d['graph'], 'trace', round(d['trace_ms'],1), 'commit', round(d['commit_ms'],1), 'iop', round(d['iop_ms'],2), 'reduction', round(d['reduction_ms'],1), 'hkzg', round(d['hyperkzg_ms'],1), 'total', round(d['total_ms'],2), d['state'])
" > $O/r06final_graphs.txt
cat $O/r06final_graphs.txt
