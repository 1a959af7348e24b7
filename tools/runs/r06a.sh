#!/bin/bash
# round 6, first call: the state at HEAD with the fused c_attn GPT-2 graph
O=gpurun_out; mkdir -p $O
{ REPS=3 ATLAS_TRACE=1 ATLAS_TRACE_ROUNDS=1 timeout 300 python tools/time_node.py; } > $O/r06a_node_rounds.txt 2>&1
timeout 600 python tools/time_graph.py node_einsum,node_relu,node_mul,nanogpt_model,gpt2_layer,gpt2 2 3 > $O/r06a_time_graph.txt 2>&1
tail -n 6 $O/r06a_time_graph.txt
timeout 600 python tools/record_device_proof.py gpt2 > $O/r06a_device_proof_gpt2.txt 2>&1; tail -n 1 $O/r06a_device_proof_gpt2.txt
timeout 300 python tools/record_device_proof.py gpt2_layer > $O/r06a_device_proof_gpt2_layer.txt 2>&1; tail -n 1 $O/r06a_device_proof_gpt2_layer.txt
