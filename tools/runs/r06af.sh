#!/bin/bash
# r06af: the sharded tests in a loop (an intermittent failure after r06ae): details of the first failure
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; O=$PWD/gpurun_out; mkdir -p $O
for i in 1 2 3 4 5 6; do
  timeout 900 python -m pytest tests/test_gpu_sharded.py -q -x -p no:cacheprovider > /tmp/sh.log 2>&1
  if grep -q "failed" /tmp/sh.log; then echo "rep $i FAILED"; grep -a "RANK_MEMORY\|AtlasError\|atlas error\|FAILED\|AssertionError\|assert " /tmp/sh.log | cut -c1-500 | head -30; break; fi
  echo "rep $i ok: $(tail -1 /tmp/sh.log)"
done > $O/r06af_sharded.txt 2>&1
cat $O/r06af_sharded.txt
