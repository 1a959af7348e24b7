timeout 1500 python -m pytest tests/test_gpu_sharded.py -q -m gpu -x 2>&1 | tail -4 | cut -c1-300
echo "--- world 1"; python tools/time_sharded.py gpt2 1 3
echo "--- world 2, members split"; python tools/time_sharded.py gpt2 2 3
echo "--- world 2, replicated"; ATLAS_REDUCTION_REPLICATED=1 python tools/time_sharded.py gpt2 2 3
echo "--- world 4, members split"; python tools/time_sharded.py gpt2 4 2
