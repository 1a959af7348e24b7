"""HyperKZG::open and commit wall clock at LOG_N with and without the fixed-base table (median of 3)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import jolt_atlas_amd as A
A.init(0)
ell = int(os.environ.get("LOG_N", "20"))
n = 1 << ell
srs = A.SRS.generate(A.random_fr(1, 1)[0], n)
p = A.MultilinearPolynomial.from_fr(A.random_fr(n, 3))
rng = np.random.default_rng(ell)
pt = [int(rng.integers(0, 1 << 62)) << 66 | int(rng.integers(0, 1 << 62)) for _ in range(ell)]


def T(f, reps=3):
    r = f(); A.sync(); ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); f(); A.sync(); ts.append(time.perf_counter() - t0)
    return round(1e3 * float(np.median(ts)), 3), r


def open_():
    return A.HyperKZG.open(srs, p, pt, A.Blake2bTranscript(b"t"))


t0, r0 = T(open_)
c0, k0 = T(lambda: srs.msm(p))
print("2^%d variable-base: open %.3f ms, commit %.3f ms" % (ell, t0, c0), flush=True)
tb = time.perf_counter()
info = srs.precompute(0, int(os.environ.get("TAB_C", "0")))
tb = time.perf_counter() - tb
t1, r1 = T(open_)
c1, k1 = T(lambda: srs.msm(p))
print("2^%d table c=%d (%.2f GB, built in %.0f ms): open %.3f ms, commit %.3f ms" % (
    ell, info["window_bits"], info["levels"] * info["n_points"] * 64 / 1e9, 1e3 * tb, t1, c1), flush=True)
assert bytes(k0) == bytes(k1)
for a, b in zip(r0, r1):
    assert np.array_equal(np.asarray(a), np.asarray(b)), "open differs"
print("outputs identical")
