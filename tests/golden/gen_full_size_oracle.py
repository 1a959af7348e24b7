#!/usr/bin/env python3
"""The oracle side of tests/test_gpu_full_size.py::test_ra_virtual_large / test_booleanity_large, computed in the build container (CPU only)
from the SAME input functions the tests use, written to tests/golden/full_size_oracle.json (sha256 of the proof rows, the challenges, the
final transcript state).  The GPU box then compares the device's proofs with these instead of spending 8-24 s of oracle time per case.

    python tests/golden/gen_full_size_oracle.py"""
import importlib.util
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("full", os.path.join(os.path.dirname(HERE), "test_gpu_full_size.py"))
T = importlib.util.module_from_spec(spec); spec.loader.exec_module(T)

doc = {}
out_path = os.path.join(HERE, "full_size_oracle.json")
if "--missing-only" in sys.argv and os.path.exists(out_path):        # keep what is there, compute the tags that are not
    doc = json.load(open(out_path))
# the lazy rounds of RaVirtual (d = 16, T >= 2^18: tests/test_gpu_full_size.py::test_ra_virtual_lazy): the all-valid chunk rows cut from the lookups
for d, log_T in [(16, 18), (16, 20)]:
    if f"ra_large2[{d}-{log_T}]" not in doc:
        doc[f"ra_large2[{d}-{log_T}]"] = T._digest(*T.ra_large_oracle(d, log_T, True))
        print("ra_large2", d, log_T, flush=True)
        json.dump(doc, open(out_path, "w"), indent=0, sort_keys=True)
for d, log_T in [(16, 18), (8, 18)]:                                  # Booleanity's lazy first rounds (test_booleanity_lazy)
    if f"bool_lazy[{d}-{log_T}]" not in doc:
        doc[f"bool_lazy[{d}-{log_T}]"] = T._digest(*T.bool_lazy_oracle(d, log_T))
        print("bool_lazy", d, log_T, flush=True)
        json.dump(doc, open(out_path, "w"), indent=0, sort_keys=True)
for d, log_T in [(4, 15), (8, 16), (16, 15), (3, 17)]:
    if f"ra_large[{d}-{log_T}]" in doc and f"ra_large2[{d}-{log_T}]" in doc:
        continue
    doc[f"ra_large[{d}-{log_T}]"] = T._digest(*T.ra_large_oracle(d, log_T, False))
    doc[f"ra_large2[{d}-{log_T}]"] = T._digest(*T.ra_large_oracle(d, log_T, True))
    print("ra_large", d, log_T, flush=True)
for d, log_T in [(8, 16), (16, 15)]:
    if f"bool_large[{d}-{log_T}]" in doc:
        continue
    doc[f"bool_large[{d}-{log_T}]"] = T._digest(*T.bool_large_oracle(d, log_T))
    print("bool_large", d, log_T, flush=True)
with open(out_path, "w") as f:
    json.dump(doc, f, indent=0, sort_keys=True)
    f.write("\n")
print("wrote", len(doc), "entries")
