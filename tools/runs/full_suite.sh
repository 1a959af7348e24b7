mkdir -p gpurun_out
TAG=${1:-r05x}
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -15 > gpurun_out/${TAG}_pytest_gpu.txt
cat gpurun_out/${TAG}_pytest_gpu.txt
