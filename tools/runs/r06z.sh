#!/bin/bash
# r06z: the dense advice polynomials committed as one batch (the i32 ones through one bucket pipeline) — parity, stage trace, A/B
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_msm.py tests/test_gpu_msm_small.py tests/test_gpu_msm_fixed_base.py tests/test_gpu_graph_golden.py tests/test_gpu_graph.py tests/test_gpu_one_element.py -q -x -p no:cacheprovider 2>&1 | tail -3 > $O/r06z_subset.txt
cat $O/r06z_subset.txt
for v in "" "ATLAS_COMMIT_NO_ALIAS=1" "" "ATLAS_COMMIT_NO_ALIAS=1"; do
  env $v timeout 300 python tools/time_graph.py nanogpt_model,gpt2 2 3 2>&1 | python -c "import sys,json
for l in sys.stdin:
    try: d=json.loads(l); print('[$v]', d['graph'], 'commit', round(d['commit_ms'],1), 'iop', round(d['iop_ms'],2), 'reduction', round(d['reduction_ms'],1), 'total', round(d['total_ms'],2), d['state'])
    except Exception: pass"
done > $O/r06z_commit_ab.txt 2>&1
cat $O/r06z_commit_ab.txt
ATLAS_TRACE=1 timeout 300 python tools/time_graph.py gpt2 2 2 2>&1 | grep -a "commit_witness" | tail -3 > $O/r06z_commit_trace.txt
cat $O/r06z_commit_trace.txt
