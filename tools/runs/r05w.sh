timeout 600 python -m pytest tests/test_gpu_ra.py tests/test_gpu_graph.py tests/test_gpu_nodes.py -q -m gpu -x 2>&1 | tail -3
for i in 1 2 3; do
ATLAS_GRAPH_VERIFY=0 python tools/time_graph.py gpt2 2 2 2>&1 | tail -1 | cut -c1-200
done
python tools/time_graph.py nanogpt_model 2 3 2>&1 | tail -1 | cut -c1-250
DIMS=16,64,1024,14 REPS=3 ATLAS_TRACE=1 ATLAS_TRACE_ROUNDS=1 python tools/time_node.py 2>&1 | grep -v "^\[atlas trace\]   round" | grep -B22 "degree 17" | tail -24
