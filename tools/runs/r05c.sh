mkdir -p gpurun_out; O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_psshout.py tests/test_gpu_shout.py tests/test_gpu_nodes.py tests/test_gpu_graph.py tests/test_gpu_graph_golden.py tests/test_gpu_softmax.py -q -m gpu -x 2>&1 | tail -4
ATLAS_PROF=1 ATLAS_GRAPH_VERIFY=0 python tools/time_graph.py gpt2 2 2 > $O/r05c_prof_gpt2.txt 2>&1
tail -1 $O/r05c_prof_gpt2.txt | cut -c1-400
ATLAS_GRAPH_VERIFY=0 ATLAS_RC_REF_PHASES=1 python tools/time_graph.py gpt2 2 2 2>&1 | tail -1 | cut -c1-400
ATLAS_GRAPH_VERIFY=0 ATLAS_SH_NO_WORDS=1 python tools/time_graph.py gpt2 2 2 2>&1 | tail -1 | cut -c1-400
python tools/time_graph.py nanogpt_model,gpt2_layer 2 3 2>&1 | tail -2 | cut -c1-400
