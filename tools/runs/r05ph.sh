timeout 900 python -m pytest tests/test_gpu_psshout.py tests/test_gpu_nodes.py tests/test_gpu_one_element.py -q -m gpu -x 2>&1 | tail -3
for i in 1 2; do
python tools/time_graph.py node_relu,nanogpt_model,microgpt_model 2 5 2>&1 | grep "^{" | python -c "
import sys,json
print('8-bit  ', ' '.join('%s %.3f/%.1f' % (json.loads(l)['graph'][:9], json.loads(l)['iop_ms'], json.loads(l)['total_ms']) for l in sys.stdin))"
ATLAS_PS_REF_PHASES=1 python tools/time_graph.py node_relu,nanogpt_model,microgpt_model 2 5 2>&1 | grep "^{" | python -c "
import sys,json
print('ref cut', ' '.join('%s %.3f/%.1f' % (json.loads(l)['graph'][:9], json.loads(l)['iop_ms'], json.loads(l)['total_ms']) for l in sys.stdin))"
done
