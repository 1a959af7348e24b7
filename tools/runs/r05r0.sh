timeout 900 python -m pytest tests/test_gpu_sumcheck.py -q -m gpu -x 2>&1 | tail -2
python bench.py --no-pmc --no-node --no-graph --no-msm --no-cpu-baseline --steps 30 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('step', round(d['ms_per_step'],4), 'frac', round(d['roofline']['frac'],3), 'pass_ms', round(d['roofline']['pass_ms'],4), d['git'])"
