#!/bin/bash
# round 6: the whole GPU suite with durations (what to buy back), after the hardening commit
mkdir -p gpurun_out
timeout 1700 python -m pytest tests -q -m gpu --durations=200 -p no:cacheprovider 2>&1 | tail -230 > gpurun_out/r06c_pytest_gpu.txt
tail -5 gpurun_out/r06c_pytest_gpu.txt
