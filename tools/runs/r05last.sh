python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" 2>&1 | tail -2
timeout 1200 python -m pytest tests/test_gpu_graph_golden.py tests/test_gpu_sharded.py tests/test_gpu_one_element.py tests/test_gpu_sumcheck.py -q -m gpu -x 2>&1 | tail -2
python bench.py --no-pmc --no-cpu-baseline --steps 20 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('step', round(d['ms_per_step'],4), 'frac', round(d['roofline']['frac'],3), 'gpt2', round(d['prove_graph']['gpt2']['prove_graph_ms'],1), d['prove_graph']['gpt2']['proof_sha16'], 'nanogpt', round(d['prove_graph']['nanogpt']['prove_graph_ms'],1), d['git'])"
