timeout 1200 python -m pytest tests/test_gpu_nodes.py tests/test_gpu_graph_golden.py tests/test_gpu_ra.py -q -m gpu -x 2>&1 | tail -3
for i in 1 2; do python tools/time_graph.py node_relu,node_add,node_mul,nanogpt_model 2 4 2>&1 | grep "^{" | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['graph'], 'total', round(d['total_ms'],2), 'iop', round(d['iop_ms'],3))"; done
echo "--- gathered rows (ATLAS_BOOL_LAZY_LOG=31)"
ATLAS_BOOL_LAZY_LOG=31 python tools/time_graph.py node_relu,node_add,node_mul,nanogpt_model 2 4 2>&1 | grep "^{" | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['graph'], 'total', round(d['total_ms'],2), 'iop', round(d['iop_ms'],3))"
