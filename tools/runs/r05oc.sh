timeout 900 python -m pytest tests/test_gpu_one_element.py -q -m gpu 2>&1 | tail -6 | cut -c1-300
