#!/usr/bin/env python3
"""Merge the JSON line tools/record_device_proof.py printed on the GPU box into tests/golden/graph_proofs.json (entry made with --trace-only)."""
import json, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "graph_proofs.json")
rec = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
doc = json.load(open(OUT))
e = doc["graphs"][rec.pop("graph")]
e.update(rec)
with open(OUT, "w") as f:
    json.dump(doc, f, indent=1); f.write("\n")
print("merged", rec)
