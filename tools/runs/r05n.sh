mkdir -p gpurun_out; O=gpurun_out
timeout 1200 python -m pytest tests/test_gpu_reduced_openings.py tests/test_gpu_opening.py tests/test_gpu_batched.py tests/test_gpu_msm.py tests/test_gpu_graph_golden.py tests/test_gpu_sharded.py tests/test_gpu_graph.py -q -m gpu -x -k "not gpt2" 2>&1 | tail -5
for i in 1 2; do
ATLAS_GRAPH_VERIFY=0 python tools/time_graph.py gpt2 2 2 2>&1 | tail -1 | cut -c1-330
ATLAS_GRAPH_VERIFY=0 ATLAS_COMMIT_OLD_CUT=1 python tools/time_graph.py gpt2 2 2 2>&1 | tail -1 | cut -c1-200
ATLAS_GRAPH_VERIFY=0 ATLAS_HOST_THREADS=12 python tools/time_graph.py gpt2 2 2 2>&1 | tail -1 | cut -c1-200
done
python tools/time_graph.py nanogpt_model,gpt2_layer 2 3 2>&1 | tail -2 | cut -c1-400
python tools/record_device_proof.py gpt2 > $O/device_proof_gpt2.json 2> $O/device_proof_gpt2.err; tail -1 $O/device_proof_gpt2.json
ATLAS_TRACE=1 ATLAS_GRAPH_VERIFY=0 timeout 300 python tools/time_graph.py gpt2 2 1 2>&1 | grep -E "prove_reduced_openings|batched_prove \(|onehot pool" | cut -c1-330 | tail -8
