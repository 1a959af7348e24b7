#!/bin/bash
# What capture_round.sh does not take: usage  tools/capture_extra.sh r04d   (through gpurun from the repo root; writes gpurun_out/<tag>_*)
#   <tag>_nanogpt_gaps.txt       idle time of the device per (kernel before -> kernel after) pair over a nanoGPT-shaped proof
#   <tag>_graph_kernel_stats.csv kernel statistics of whole proofs (nanoGPT-shaped + one GPT-2 layer)
#   <tag>_reduction_trace.txt    ATLAS_TRACE stages of prove_reduced_openings (nanoGPT- and GPT-2-shaped)
#   <tag>_pass_pmc_sq.txt / <tag>_msm_pmc_sq.txt   SQ counters of the sumcheck data passes / the MSM kernels (separate --pmc runs, --kernel-trace only)
#   <tag>_exp_collective.txt     the cost of a 64-byte exchange: board / gloo / RCCL floor
TAG=${1:-rXX}
cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
R=$PWD
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_g_$TAG -o r -- python $R/tools/time_graph.py nanogpt_model,gpt2_layer 2 2 > /tmp/prof_g_$TAG.log 2>&1 )
DB=$(find /tmp/prof_g_$TAG -name "*.db" | head -1)
python tools/rocprof_summary.py $DB $O/${TAG}_graph_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python tools/time_graph.py nanogpt_model,gpt2_layer 2 2" > /dev/null
( cd /tmp && timeout 300 rocprofv3 --kernel-trace -d /tmp/prof_n_$TAG -o r -- python $R/tools/time_graph.py nanogpt_model 2 2 > /tmp/prof_n_$TAG.log 2>&1 )
DB=$(find /tmp/prof_n_$TAG -name "*.db" | head -1)
python tools/rocprof_gaps.py $DB 480 60 > $O/${TAG}_nanogpt_gaps.txt 2>&1
ATLAS_TRACE=1 timeout 300 python tools/time_graph.py nanogpt_model,gpt2 2 1 2>&1 | grep -E "prove_reduced_openings|batched_prove \(|onehot pool|^\{" | cut -c1-260 > $O/${TAG}_reduction_trace.txt
SQ="SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE"
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $SQ -d /tmp/prof_sq1_$TAG -o r -- python $R/bench.py --fs device --no-pmc --no-node --no-graph --no-msm --no-cpu-baseline --steps 3 > /tmp/prof_sq1_$TAG.log 2>&1 )
DB=$(find /tmp/prof_sq1_$TAG -name "*.db" | head -1)
python tools/pmc_sq_summary.py $DB $O/${TAG}_pass_pmc_sq.txt k_dot_ > /dev/null 2>&1
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $SQ -d /tmp/prof_sq2_$TAG -o r -- python $R/bench.py --no-pmc --no-node --no-graph --no-cpu-baseline --steps 2 --warmup 1 > /tmp/prof_sq2_$TAG.log 2>&1 )
DB=$(find /tmp/prof_sq2_$TAG -name "*.db" | head -1)
python tools/pmc_sq_summary.py $DB $O/${TAG}_msm_pmc_sq.txt k_msm_ k_tab_ > /dev/null 2>&1
timeout 200 python tools/exp_collective.py > $O/${TAG}_exp_collective.txt 2>&1
tail -n 4 $O/${TAG}_reduction_trace.txt; head -12 $O/${TAG}_nanogpt_gaps.txt; head -6 $O/${TAG}_pass_pmc_sq.txt; cat $O/${TAG}_exp_collective.txt | cut -c1-600
