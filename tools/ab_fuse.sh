for rep in 1 2 3; do
for cfg in both ra_only none; do
  unset ATLAS_RA_NO_FUSE ATLAS_BOOL_NO_FUSE
  if [ $cfg = ra_only ]; then export ATLAS_BOOL_NO_FUSE=1; fi
  if [ $cfg = none ]; then export ATLAS_RA_NO_FUSE=1; fi
  echo -n "$cfg "; ATLAS_GRAPH_VERIFY=0 timeout 200 python tools/time_graph.py nanogpt 2 4 2>&1 | grep total_ms | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(round(d['iop_ms'],1), round(d['total_ms'],1))"
done; done
