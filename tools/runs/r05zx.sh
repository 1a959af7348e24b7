echo "--- nanogpt world 4 split"; python tools/time_sharded.py nanogpt_model 4 2 2>&1 | cut -c1-250
echo "--- nanogpt world 4 replicated"; ATLAS_REDUCTION_REPLICATED=1 python tools/time_sharded.py nanogpt_model 4 2 2>&1 | cut -c1-250
echo "--- nanogpt world 1"; python tools/time_sharded.py nanogpt_model 1 2 2>&1 | cut -c1-250
echo "--- gpt2 world 4 replicated"; ATLAS_REDUCTION_REPLICATED=1 python tools/time_sharded.py gpt2 4 1 2>&1 | grep -E "RANK|AtlasError" | cut -c1-250
