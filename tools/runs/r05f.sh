mkdir -p gpurun_out; O=gpurun_out
ATLAS_PROF=1 ATLAS_GRAPH_VERIFY=0 python tools/time_graph.py gpt2 2 2 > $O/r05f_prof_gpt2.txt 2>&1
tail -1 $O/r05f_prof_gpt2.txt | cut -c1-300
python tools/gpt2_by_operator.py gpt2 2 > $O/r05f_gpt2_by_operator.txt 2>&1
