mkdir -p gpurun_out; O=gpurun_out
timeout 1200 python -m pytest tests/test_gpu_ra.py tests/test_gpu_nodes.py tests/test_gpu_graph_golden.py tests/test_gpu_lane_stress.py tests/test_gpu_batched.py tests/test_gpu_graph.py tests/test_gpu_leaks.py -q -m gpu -x 2>&1 | tail -6
for i in 1 2; do
ATLAS_GRAPH_VERIFY=0 python tools/time_graph.py gpt2 2 2 2>&1 | tail -1 | cut -c1-330
ATLAS_GRAPH_VERIFY=0 ATLAS_NO_LANE_BUILD=1 python tools/time_graph.py gpt2 2 2 2>&1 | tail -1 | cut -c1-330
done
python tools/time_graph.py nanogpt_model,gpt2_layer 2 3 2>&1 | tail -2 | cut -c1-400
ATLAS_GRAPH_TRACE=1 ATLAS_GRAPH_VERIFY=0 python tools/time_graph.py nanogpt_model 2 3 2>&1 | grep "device pool"
