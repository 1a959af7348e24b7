timeout 1200 python -m pytest tests/test_gpu_graph.py tests/test_gpu_one_element.py tests/test_gpu_graph_fuzz.py -q -m gpu -x 2>&1 | tail -40 | cut -c1-300
