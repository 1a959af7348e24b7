timeout 1200 python -m pytest tests/test_gpu_graph_golden.py tests/test_gpu_full_size.py -q -m gpu -x 2>&1 | tail -3
ATLAS_BOOL_LAZY_LOG=15 timeout 900 python -m pytest tests/test_gpu_nodes.py tests/test_gpu_graph.py -q -m gpu -x 2>&1 | tail -2
