for i in 1 2 3; do timeout 300 python -m pytest "tests/test_gpu_graph.py::test_graph_proof_matches_oracle" -q -m gpu -x -k trig 2>&1 | tail -1; done
timeout 900 python -m pytest tests/test_gpu_graph_golden.py -q -m gpu -x -k gpt2_12 --durations=3 2>&1 | tail -8
timeout 1200 python -m pytest tests/test_gpu_sharded.py -q -m gpu -x -k "gpt2" --durations=3 2>&1 | tail -8
ATLAS_GRAPH_VERIFY=0 python tools/time_graph.py gpt2 2 2 2>&1 | tail -1 | cut -c1-330
