timeout 1700 python tools/stress_proofs.py gpt2:25,nanogpt_model:80,gpt2_layer:80,microgpt_model:300 2>&1 | tail -8
