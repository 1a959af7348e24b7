"""Scaling check at 2^24 coefficients (GPT-2's joint polynomial): fused sumcheck, dense opening, evaluate, RLC."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import jolt_atlas_amd as A
from jolt_atlas_amd import instances as I, rlc
A.init(0)
n = int(os.environ.get("LOG_N", "24"))
L = A.MultilinearPolynomial.from_fr(A.random_fr(1 << n, 1)); R = A.MultilinearPolynomial.from_fr(A.random_fr(1 << n, 2))
def T(f, reps=3):
    f(); A.sync(); ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); f(); A.sync(); ts.append(time.perf_counter() - t0)
    return round(1e3 * float(np.median(ts)), 3)
claim = A.random_fr(1, 3)[0]
def sc():
    p = A.EinsumDotProver(L.clone(), R.clone(), None, 0, 0, 0)
    A.Sumcheck.prove(p, claim, A.Blake2bTranscript(b"t"), n); p.free()
def clone_only():
    a, b = L.clone(), R.clone(); a.free(); b.free()
print("clone x2 ms", T(clone_only))
print("fused sumcheck 2^%d ms (incl. the two clones)" % n, T(sc))
pt = A.random_fr(n, 4)
def dense():
    i = I.dense_opening(L.clone(), pt); i.prove(claim, A.Blake2bTranscript(b"t")); i.free()
print("dense opening ms (incl. clone)", T(dense))
print("evaluate ms", T(lambda: A.evaluate(L, pt)))
co = A.random_fr(2, 5)
def do_rlc():
    j = rlc.build_materialized_rlc([(L, co[0]), (R, co[1])], []); j.free()
print("rlc 2 dense ms", T(do_rlc))
