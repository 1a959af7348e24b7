T="tests/test_gpu_graph.py::test_graph_proof_matches_oracle"
for e in "X=1" "ATLAS_PS_NO_TAIL=1" "ATLAS_NO_TAGGED_ROWS=1" "ATLAS_PS_NO_SIGN=1"; do
  echo "== $e"; env $e timeout 300 python -m pytest "$T" -q -m gpu -x -k trig 2>&1 | tail -2
done
