run() { python tools/time_graph.py node_add,node_mul,nanogpt_model 2 5 2>&1 | grep "^{" | python -c "
import sys,json
print('$1', ' '.join('%s %.3f/%.1f' % (json.loads(l)['graph'][:9], json.loads(l)['iop_ms'], json.loads(l)['total_ms']) for l in sys.stdin))"; }
for rep in 1 2 3; do
run base
ATLAS_RA_LAZY_LOG=16 run ralazy16
ATLAS_RA_LAZY_LOG=16 ATLAS_BOOL_LAZY_LOG=16 run bothlazy16
done
