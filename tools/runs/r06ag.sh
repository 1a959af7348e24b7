#!/bin/bash
# r06ag: runtimes of one process that share a device keep their lanes on the library stream — the one-process tests in a loop (30 runs)
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; O=$PWD/gpurun_out; mkdir -p $O
for i in $(seq 1 30); do
  timeout 300 python -m pytest tests/test_gpu_sharded.py -q -x -vv -k "one_process" -p no:cacheprovider > /tmp/op.log 2>&1
  if grep -q "failed" /tmp/op.log; then grep -a "AtlasError\|atlas error" /tmp/op.log | cut -c1-400 | head -6; echo "rep $i FAILED"; break; fi
  echo "rep $i ok: $(tail -1 /tmp/op.log)"
done > $O/r06ag_one_process.txt 2>&1
tail -4 $O/r06ag_one_process.txt
