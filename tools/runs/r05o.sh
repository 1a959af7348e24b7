T="tests/test_gpu_graph.py::test_graph_proof_matches_oracle"
for e in "X=1" "ATLAS_PS_NO_TAIL=1"; do
  echo "== $e"; env $e ATLAS_GRAPH_TRACE=2 timeout 300 python -m pytest "$T" -q -m gpu -x -k trig 2>&1 | grep -E "atlas error|passed|failed|atlas graph" | tail -12
done
