ATLAS_TRACE=0 timeout 900 python -m pytest tests/test_gpu_full_size.py -q -m gpu -x -k "lazy" 2>&1 | tail -5 | cut -c1-300
