export ATLAS_BENCH_BACKEND=gloo
timeout 1200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 3 --warmup 1 > gpurun_out/r05final_bench_gpus2_gloo_smoke.json 2> gpurun_out/r05final_bench_gpus2.err
tail -c 1500 gpurun_out/r05final_bench_gpus2_gloo_smoke.json; tail -5 gpurun_out/r05final_bench_gpus2.err | cut -c1-300
