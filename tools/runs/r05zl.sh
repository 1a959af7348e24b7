timeout 900 python -m pytest tests/test_gpu_sharded.py -q -m gpu -x -k "one_process" 2>&1 | tail -3
bash tools/runs/r05zk.sh
