timeout 1200 python -m pytest tests/test_gpu_sharded.py -q -m gpu -x 2>&1 | tail -12 | cut -c1-400
ATLAS_GRAPH_VERIFY=0 python tools/time_graph.py gpt2,nanogpt_model 2 3 2>&1 | grep "^{" | cut -c1-200
