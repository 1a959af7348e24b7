timeout 900 python -m pytest tests/test_gpu_one_element.py -q -m gpu  2>&1 | grep -E "^E  |Error|passed|failed" | head -40
timeout 900 python -m pytest tests/test_gpu_graph_fuzz.py tests/test_gpu_graph.py -q -m gpu -x 2>&1 | tail -3
