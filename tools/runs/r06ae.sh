#!/bin/bash
# r06ae: the one-process two-thread proof failed 1 run in 6 with the evaluation on its own stream (a cross-stream event wait among another runtime's polling
# launches): the side path is now off while another runtime drives the device — twelve repetitions, then the sharded tests
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; O=$PWD/gpurun_out; mkdir -p $O
for i in 1 2 3 4 5 6 7 8 9 10 11 12; do
  timeout 300 python -m pytest tests/test_gpu_sharded.py -q -x -vv -k "one_process" -p no:cacheprovider > /tmp/op.log 2>&1
  if grep -q "failed" /tmp/op.log; then grep -a "AtlasError\|atlas error" /tmp/op.log | cut -c1-400 | head -6; echo "rep $i FAILED"; break; fi
  echo "rep $i ok: $(tail -1 /tmp/op.log)"
done > $O/r06ae_one_process.txt 2>&1
cat $O/r06ae_one_process.txt
timeout 900 python -m pytest tests/test_gpu_sharded.py tests/test_gpu_graph_golden.py tests/test_gpu_hardening.py tests/test_gpu_lifecycle.py -q -x -p no:cacheprovider 2>&1 | tail -2 | tee -a $O/r06ae_one_process.txt
