"""Per-round latency of the element-wise operator provers at small T: pipelined (round channel) vs host-stepped."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import jolt_atlas_amd as A
from jolt_atlas_amd import instances as I
A.init(0)
OPS = {"add": (0, 2), "mul": (5, 2), "cube": (6, 1), "div": (7, 4), "dot": (9, 2)}
def T(mk, n, reps=8):
    ts = []
    for _ in range(reps):
        inst = mk(); A.sync()
        t0 = time.perf_counter(); inst.prove(np.zeros(4, dtype=np.uint64), A.Blake2bTranscript(b"t")); ts.append(time.perf_counter() - t0)
        inst.free()
    return 1e6 * float(np.median(ts[2:])) / n
for n in (int(x) for x in os.environ.get("SIZES", "6,10,14,18,20").split(",")):
    rn = A.random_fr(n, 2)
    for name, (code, k) in OPS.items():
        polys = [A.MultilinearPolynomial.from_fr(A.random_fr(1 << n, 10 + i)) for i in range(k)]
        mk = lambda: I.elementwise(code, polys, rn, None)
        os.environ.pop("ATLAS_NO_PIPELINE", None)
        a = T(mk, n)
        os.environ["ATLAS_NO_PIPELINE"] = "1"
        b = T(mk, n)
        os.environ.pop("ATLAS_NO_PIPELINE", None)
        print(f"elementwise {name:5s} n={n:2d}: pipelined {a:6.2f} us/round, host-stepped {b:6.2f} us/round", flush=True)
        for p in polys: p.free()
