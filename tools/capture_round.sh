#!/bin/bash
# One GPU call that produces everything profiles/ holds for a round: usage  tools/capture_round.sh r02b [skip-tests]
# (run through gpurun from the repo root; writes gpurun_out/<tag>_*).
TAG=${1:-rXX}
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
if [ "$2" != "skip-tests" ]; then
  timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -5 > $O/${TAG}_pytest_gpu.txt
  cat $O/${TAG}_pytest_gpu.txt
fi
timeout 600 python bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err; tail -c 600 $O/${TAG}_bench.json
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -o r -- python $OLDPWD/bench.py --no-pmc --no-node --steps 10 > /tmp/prof_$TAG.log 2>&1 )
DB=$(find /tmp/prof_$TAG -name "*.db" | head -1)
python tools/rocprof_summary.py $DB $O/${TAG}_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python bench.py --no-pmc --no-node --steps 10" > /dev/null
python tools/rocprof_dispatch_csv.py $DB 22 $O/${TAG}_dispatches.csv "per-dispatch data passes of the 2^22 instances of the same run"
{ timeout 300 python tools/time_small.py; timeout 300 python tools/time_small_ew.py; } > $O/${TAG}_small_instances.txt 2>&1
timeout 600 python tools/time_components.py > $O/${TAG}_components.json 2> $O/${TAG}_components.err
for l in 20 22 24; do LOG_N=$l timeout 300 python tools/time_open_tab.py; done > $O/${TAG}_hyperkzg_open.txt 2>&1
{ REPS=5 ATLAS_TRACE=1 timeout 300 python tools/time_node.py; } > $O/${TAG}_node_einsum.txt 2>&1
{ for l in 20 22 24; do LOG_N=$l TAB_C=0 timeout 300 python tools/time_msm_tab.py; done; timeout 300 python tools/time_msm_skew.py; } > $O/${TAG}_msm.txt 2>&1
for x in exp_channel exp_channel2 exp_hostread; do
  timeout 200 /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/$x.hip -o /tmp/$x 2> /tmp/$x.build.log && timeout 120 /tmp/$x > $O/r02_$x.txt 2>&1
done
for f in small_instances hyperkzg_open node_einsum; do tail -n 3 $O/${TAG}_$f.txt; done
