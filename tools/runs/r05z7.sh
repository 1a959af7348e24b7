timeout 900 python -m pytest tests/test_gpu_softmax.py tests/test_gpu_graph.py tests/test_gpu_graph_golden.py tests/test_gpu_graph_fuzz.py -q -m gpu -x 2>&1 | tail -3
for i in 1 2; do
ATLAS_GRAPH_VERIFY=0 python tools/time_graph.py gpt2 2 2 2>&1 | tail -1 | cut -c1-200
ATLAS_GRAPH_VERIFY=0 ATLAS_SM_DEVICE_RAF=1 python tools/time_graph.py gpt2 2 2 2>&1 | tail -1 | cut -c1-200
done
python tools/time_graph.py nanogpt_model 2 3 2>&1 | tail -1 | cut -c1-200
ATLAS_SM_DEVICE_RAF=1 python tools/time_graph.py nanogpt_model 2 3 2>&1 | tail -1 | cut -c1-200
