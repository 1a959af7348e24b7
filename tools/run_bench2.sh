cd $GRAFT_REPO_ROOT
ATLAS_BENCH_BACKEND=gloo timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29871 bench.py --gpus 2 --steps 5 --warmup 2 --n-vars 18 2>&1 | tail -3
