#!/bin/bash
# Bisection of the lane-stream nondeterminism: tools/stress_lanes.py under one knob at a time (run through gpurun; writes gpurun_out/<tag>_bisect_lanes.txt)
TAG=${1:-rXX}; REPS=${2:-1500}
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
O=gpurun_out/${TAG}_bisect_lanes.txt; : > $O
run() { echo "== $*" >> $O; ( export ATLAS_STRESS_PRE_STREAMS=1 "$@"; timeout 600 python tools/stress_lanes.py $REPS 2>&1 | tail -12 ) >> $O; }
if [ "$3" = "nogate" ]; then      # the pre-gate state: which ingredient does the nondeterminism need?
  run ATLAS_LANE_NO_GATE=1
  run ATLAS_LANE_NO_GATE=1 ATLAS_LANE_EVENTS=1
  run ATLAS_LANE_NO_GATE=1 ATLAS_LANE_EVENTS=1 ATLAS_NO_MAIL_TAIL=1
  run ATLAS_LANE_NO_GATE=1 ATLAS_LANE_EVENTS=1 ATLAS_CH_HOST_POLL=1
  run ATLAS_LANE_NO_GATE=1 ATLAS_LANE_EVENTS=1 ATLAS_LANE_ONE_STREAM=1
  run ATLAS_LANE_NO_GATE=1 ATLAS_LANE_EVENTS=1 ATLAS_NO_POOL=1
  run ATLAS_LANE_EVENTS=1
  cat $O; exit 0
fi
run X=1
run X=2
run ATLAS_NO_LANE_STREAMS=1
run ATLAS_LANE_ONE_STREAM=1
run ATLAS_NO_MAIL_TAIL=1
run ATLAS_CH_HOST_POLL=1
run ATLAS_NO_POOL=1
cat $O
