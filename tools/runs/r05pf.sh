timeout 1200 python -m pytest tests/test_gpu_nodes.py tests/test_gpu_graph.py tests/test_gpu_graph_golden.py tests/test_gpu_psshout.py -q -m gpu -x 2>&1 | tail -3
for i in 1 2 3; do
python tools/time_graph.py node_relu,node_add,node_mul,nanogpt_model,gpt2 2 4 2>&1 | grep "^{" | python -c "
import sys,json
print('prefetch', ' '.join('%s %.3f/%.1f' % (json.loads(l)['graph'][:9], json.loads(l)['iop_ms'], json.loads(l)['total_ms']) for l in sys.stdin))"
ATLAS_PS_NO_PREFETCH=1 python tools/time_graph.py node_relu,node_add,node_mul,nanogpt_model,gpt2 2 4 2>&1 | grep "^{" | python -c "
import sys,json
print('no pref ', ' '.join('%s %.3f/%.1f' % (json.loads(l)['graph'][:9], json.loads(l)['iop_ms'], json.loads(l)['total_ms']) for l in sys.stdin))"
done
