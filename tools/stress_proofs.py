"""Repeat whole proofs and hold every repetition to the first one's bytes (a latent race in a kernel or in the round channel shows up as a
different proof or as 'no answer from the device'):  python tools/stress_proofs.py gpt2:12,nanogpt_model:40,microgpt_model:150"""
import hashlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import build_graphs as BG
import jolt_atlas_amd as A
from jolt_atlas_amd import graph as GG
A.init(0)
for item in (sys.argv[1] if len(sys.argv) > 1 else "nanogpt_model:20").split(","):
    name, reps = item.split(":"); reps = int(reps)
    nodes, outputs, inputs = getattr(BG, name)()
    nv = BG.max_vars(nodes)
    srs = A.SRS.generate(np.array([0x1234567, 0, 0, 0], dtype=np.uint64), 1 << nv)
    if nv >= 16: srs.precompute()
    G = GG.Graph(nodes, outputs)
    first, ts = None, []
    for rep in range(reps):
        t0 = time.time()
        proof, state, tm = G.prove(srs, inputs)
        ts.append(tm["total_ms"])
        h = hashlib.sha256(proof).hexdigest()
        if first is None: first = h
        assert h == first, "repetition %d of %s: proof %s != %s" % (rep, name, h[:16], first[:16])
    print("%s: %d proofs, all %s; total_ms min %.1f median %.1f max %.1f" % (name, reps, first[:16], min(ts), float(np.median(ts)), max(ts)), flush=True)
    G.free(); srs.free()
