import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import jolt_atlas_amd as A
from jolt_atlas_amd import instances as I, reduced, rlc
A.init(0)
rng = np.random.default_rng(1)
srs20 = A.SRS.generate(A.random_fr(1, 1)[0], 1 << 20)
P = A.MultilinearPolynomial.from_fr(A.random_fr(1 << 20, 32))
pt = [int.from_bytes(rng.bytes(16), "little") for _ in range(20)]
def T(f, n=3):
    f(); A.sync(); ts = []
    for _ in range(n):
        t0 = time.perf_counter(); f(); A.sync(); ts.append(time.perf_counter() - t0)
    return 1e3 * float(np.median(ts))
print("hyperkzg_open 2^20 ms", T(lambda: A.HyperKZG.open(srs20, P, pt, A.Blake2bTranscript(b"t"))))
print("msm 2^20 ms", T(lambda: srs20.msm(P)))
rcy = A.random_fr(16, 31)
ks = [rng.integers(0, 16, size=1 << 16, dtype=np.int32) for _ in range(16)]
ras = [A.random_fr(4, 35 + q) for q in range(16)]
def batched():
    gi = [I.dense_opening(P.clone(), A.random_fr(20, 33))]
    gi += [I.onehot_opening(ks[q], 4, ras[q], rcy) for q in range(16)]
    A.BatchedSumcheck.prove(gi, [np.zeros(4, dtype=np.uint64)] * 17, A.Blake2bTranscript(b"t"))
    for x in gi: x.free()
print("batched ungrouped ms", T(batched))
co = A.random_fr(17, 5)
def do_rlc():
    j = rlc.build_materialized_rlc([(P, co[0])], [(ks[q], 16, co[1 + q]) for q in range(16)]); j.free()
print("rlc ms", T(do_rlc))
ops = [dict(poly=P, point=A.random_fr(20, 33), claim=A.random_fr(1, 34)[0])]
for q in range(16):
    ops.append(dict(k=ks[q], log_K=4, r_address=ras[q], r_cycle=rcy, claim=A.random_fr(1, 60 + q)[0]))
print("prove_reduced_openings ms", T(lambda: reduced.prove_reduced_openings(ops, srs20, A.Blake2bTranscript(b"t"))))
