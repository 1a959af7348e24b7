"""Per-round host timing of ONE 64-bit saturating-clamp read-raf (the lookup behind every Add / fused rescale) and ONE one-hot batch at
log_T = LOG_T (default 14): ATLAS_TRACE=1 ATLAS_TRACE_ROUNDS=1 python tools/time_clamp_rounds.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import jolt_atlas_amd as A
from jolt_atlas_amd import instances as I
A.init(0)
n = int(os.environ.get("LOG_T", "14"))
rng = np.random.default_rng(3)
vals = rng.integers(-(1 << 20), 1 << 20, size=1 << n, dtype=np.int64).astype(np.uint64)   # sign-extended 64-bit indices
rc = A.random_fr(n, 4); gamma = A.random_fr(1, 7)[0]
for rep in range(3):
    inst = I.ps_shout_clamp(vals, 64, 31, 1, rc, gamma); A.sync()
    if rep == 2: os.environ["ATLAS_TRACE_GO"] = "1"
    t0 = time.perf_counter(); inst.prove(np.zeros(4, dtype=np.uint64), A.Blake2bTranscript(b"t")); dt = time.perf_counter() - t0
    inst.free()
    print("clamp64 log_T=%d: %.3f ms" % (n, 1e3 * dt), flush=True)
