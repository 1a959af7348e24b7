timeout 900 python -m pytest tests/test_gpu_nodes.py tests/test_gpu_graph.py tests/test_gpu_graph_golden.py tests/test_gpu_one_element.py -q -m gpu -x 2>&1 | tail -3
python tools/time_graph.py node_relu,node_einsum,node_mul,microgpt_model 2 4 2>&1 | tail -4 | cut -c1-220
