import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import jolt_atlas_amd as A
from jolt_atlas_amd import instances as I
A.init(0)
rng = np.random.default_rng(1)
def T(mk, n=5):
    ts = []
    for _ in range(n + 1):
        inst = mk(); A.sync()
        t0 = time.perf_counter(); inst.prove(np.zeros(4, dtype=np.uint64), A.Blake2bTranscript(b"t")); ts.append(time.perf_counter() - t0)
        inst.free()
    return 1e3 * float(np.median(ts[1:]))
for n in (6, 10, 14):
    pol = A.random_fr(1 << n, 1); pt = A.random_fr(n, 2)
    print("dense_opening", n, T(lambda: I.dense_opening(A.MultilinearPolynomial.from_fr(pol), pt)), "ms")
    H = [rng.integers(0, 16, size=1 << n, dtype=np.int32) for _ in range(16)]
    ch = A.random_fr(64, 3).reshape(16, 4, 4); rc = A.random_fr(n, 4)
    print("ra_virtual d16", n, T(lambda: I.ra_virtual(H, 4, ch, rc)), "ms")
    print("booleanity d16", n, T(lambda: I.booleanity(np.zeros((16, 16, 4), dtype=np.uint64), H, 4, A.random_fr(16, 5), A.random_fr(4, 6), rc)), "ms")
    act = (rng.integers(-(1 << 14), 1 << 14, size=1 << n, dtype=np.int64) & 0xffffffff).astype(np.uint64)
    print("ps_shout relu32", n, T(lambda: I.ps_shout_relu(act, 32, rc, A.random_fr(1, 7)[0])), "ms")
    # fused single-instance dot for comparison
    def dot():
        p = A.EinsumDotProver(A.MultilinearPolynomial.from_fr(pol), A.MultilinearPolynomial.from_fr(pol))
        class W:
            def prove(self, c, t): A.Sumcheck.prove(p, c, t, n)
            def free(self): p.free()
        return W()
    print("fused dot", n, T(dot), "ms")
