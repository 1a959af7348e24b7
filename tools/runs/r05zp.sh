run() { ATLAS_GRAPH_VERIFY=0 python tools/time_graph.py gpt2 2 3 2>&1 | grep "^{" | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('$1', d['graph'], round(d['total_ms'],1), 'iop', round(d['iop_ms'],1))"; }
for rep in 1 2; do
run base
ATLAS_BOOL_SPLIT_LOG=14 run bool14
ATLAS_BOOL_SPLIT_LOG=15 run bool15
ATLAS_BOOL_SPLIT_LOG=16 run bool16
ATLAS_RA_SPLIT_MIN=13 run rasplit13
ATLAS_RA_SPLIT_MIN=11 run rasplit11
done
