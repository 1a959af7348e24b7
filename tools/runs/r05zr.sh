ATLAS_TRACE=1 ATLAS_TRACE_ROUNDS=1 python tools/time_graph.py node_relu 2 2 > gpurun_out/r05zr_relu_rounds.txt 2>&1
wc -l gpurun_out/r05zr_relu_rounds.txt
