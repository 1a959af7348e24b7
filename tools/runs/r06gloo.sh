#!/bin/bash
# r06gloo: bench.py --gpus 2 under torch.distributed.run with gloo on the ONE GPU of the box (both ranks share it: a smoke run of the N > 1 legs, not a measurement)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
export ATLAS_BENCH_BACKEND=gloo
timeout 1200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 3 --warmup 1 > gpurun_out/r06final_bench_gpus2_gloo_smoke.json 2> gpurun_out/r06final_bench_gpus2.err
tail -c 1800 gpurun_out/r06final_bench_gpus2_gloo_smoke.json; tail -5 gpurun_out/r06final_bench_gpus2.err | cut -c1-300
