"""Fixed-base MSM wall clock at LOG_N (default 22): variable-base, then the table for each window width in
TAB_C (comma list, default "16,18,20,22") — median of 7, plus the table build time and the stage split
(sort / accumulate+reduce / fold) from the library's timing events."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import jolt_atlas_amd as A
A.init(0)
ln = int(os.environ.get("LOG_N", "22"))
srs = A.SRS.generate(A.random_fr(1, 1)[0], 1 << ln)
p = A.MultilinearPolynomial.from_fr(A.random_fr(1 << ln, 3))


def run(tag):
    ref = srs.msm(p); A.sync()
    ts = []
    for _ in range(7):
        t0 = time.perf_counter(); srs.msm(p); ts.append(time.perf_counter() - t0)
    A.set_timing(True)
    srs.msm(p)
    t = A.last_timing()
    A.set_timing(False)
    print("msm 2^%d %-22s %8.3f ms   events: total %.3f = accumulate+reduce %.3f + sort+fold %.3f" % (
        ln, tag, 1e3 * float(np.median(ts)), t.total_ms, t.pass_ms, t.fs_ms), flush=True)
    return ref


os.environ["ATLAS_MSM_TAB"] = "0"
ref = run("variable-base")
del os.environ["ATLAS_MSM_TAB"]
for c in [int(x) for x in os.environ.get("TAB_C", "16,18,20,22").split(",")]:
    t0 = time.perf_counter()
    info = srs.precompute(0, c)
    tb = time.perf_counter() - t0
    got = run("table c=%d (%d lv)" % (c, info["levels"]))
    assert bytes(got) == bytes(ref), "table path disagrees with the variable-base path"
    print("   table build %.1f ms, %.2f GB" % (1e3 * tb, info["levels"] * info["n_points"] * 64 / 1e9), flush=True)
