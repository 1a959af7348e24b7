timeout 900 python -m pytest tests/test_gpu_one_element.py -q -m gpu -k "softmax" 2>&1 | grep -E "^E  |Error|passed|failed" | head -40
