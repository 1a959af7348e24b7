timeout 900 python -m pytest tests/test_gpu_reduced_openings.py tests/test_gpu_graph_golden.py tests/test_gpu_batched.py -q -m gpu -x 2>&1 | tail -3
for i in 1 2 3; do ATLAS_GRAPH_VERIFY=0 python tools/time_graph.py gpt2 2 2 2>&1 | tail -1 | cut -c1-200; done
