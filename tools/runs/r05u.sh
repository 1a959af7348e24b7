mkdir -p gpurun_out; O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_graph.py tests/test_gpu_graph_golden.py tests/test_gpu_ra.py -q -m gpu -x 2>&1 | tail -3
python tools/gpt2_by_operator.py gpt2 2 > $O/r05u_gpt2_by_operator.txt 2>&1; head -24 $O/r05u_gpt2_by_operator.txt
ATLAS_PROF=1 ATLAS_GRAPH_VERIFY=0 python tools/time_graph.py gpt2 2 2 > $O/r05u_prof_gpt2.txt 2>&1
tail -1 $O/r05u_prof_gpt2.txt | cut -c1-250
