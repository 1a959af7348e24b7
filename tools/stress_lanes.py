"""Stress of the lane streams of a pipelined batched proof (csrc/batched.hip Pipeline): the ReLU and Einsum nodes' one-hot checks run
[RaVirtual, HammingWeight, Booleanity] on three lane streams.  Repeats the node proofs and counts distinct transcript states; run under
  ATLAS_LANE_EVENTS=1       lanes ordered behind the library stream by events (the pre-4a34137 design) instead of host waits
  ATLAS_POOL_ANYSTREAM=1    the device pool reuses a cached block whatever stream it was freed under (the pre-4a34137 rule)
  ATLAS_NO_POOL=1           no caching allocator
to see which ingredient makes a proof non-deterministic.  Between repetitions other legs run (an MSM, a dot sumcheck) so that the
stream / hardware-queue assignment looks like bench.py's."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import jolt_atlas_amd as A  # noqa: E402
from jolt_atlas_amd import node as NODE  # noqa: E402

pre = int(os.environ.get("ATLAS_STRESS_PRE_STREAMS", "0"))      # streams created BEFORE the library's: shifts the stream -> hardware queue assignment
if pre:
    # through the HIP runtime the library itself is linked against (a second runtime in the process — torch's bundled one — cannot open the
    # device once another has: "no ROCm-capable device is detected")
    import ctypes
    _hip = ctypes.CDLL("libamdhip64.so.7")      # already mapped: jolt_atlas_amd loaded libatlas_hip.so, which links it
    _keep = []
    for _ in range(pre):
        st_ = ctypes.c_void_p()
        assert _hip.hipStreamCreateWithFlags(ctypes.byref(st_), ctypes.c_uint(1)) == 0      # hipStreamNonBlocking
        buf_ = ctypes.c_void_p()
        assert _hip.hipMalloc(ctypes.byref(buf_), ctypes.c_size_t(4096)) == 0
        assert _hip.hipMemsetAsync(buf_, ctypes.c_int(0), ctypes.c_size_t(4096), st_) == 0  # some work, so that the stream owns a hardware queue
        _keep.append((st_, buf_))
    assert _hip.hipDeviceSynchronize() == 0
A.init(0)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(14)
tX = A.TensorI32(rng.integers(-(1 << 14), 1 << 14, size=1 << 16, dtype=np.int64).astype(np.int32))
tA = A.TensorI32(rng.integers(-(1 << 14), 1 << 14, size=(16, 1024), dtype=np.int64).astype(np.int32))
tB = A.TensorI32(rng.integers(-(1 << 14), 1 << 14, size=(1024, 4096), dtype=np.int64).astype(np.int32))
r0 = A.random_fr(16, 0xE1)
r1 = A.random_fr(14, 0xE2)
srs = A.SRS.generate(np.array([5, 0, 0, 0], dtype=np.uint64), 1 << 18)
sc = A.MultilinearPolynomial.from_fr(A.random_fr(1 << 18, 3))
L, R = A.random_fr(1 << 18, 1), A.random_fr(1 << 18, 2)
states_r, states_e = {}, {}
first = {}


def diff_report(kind, rep, proofs, claims):
    """where a proof departs from the first one seen: proof index, first differing 32-byte word (8-byte header words count), first differing claim"""
    p0, c0 = first.setdefault(kind, (proofs, claims))
    if p0 is proofs:
        return
    for i, (a, b) in enumerate(zip(p0, proofs)):
        if a != b:
            w = next(k for k in range(0, max(len(a), len(b)), 8) if a[k:k + 8] != b[k:k + 8])
            print(f"DIFF {kind} rep {rep}: proof {i} of {len(proofs)} (len {len(a)} / {len(b)}) first differing byte {w}", flush=True)
            break
    else:
        print(f"DIFF {kind} rep {rep}: proofs equal", flush=True)
    d = [k for k in range(min(len(c0), len(claims))) if not np.array_equal(c0[k], claims[k])]
    print(f"     claims: {len(d)} of {len(claims)} differ, first {d[:4]}", flush=True)


for rep in range(reps):
    if rep % 3 == 0:
        srs.msm_poly(sc) if hasattr(srs, "msm_poly") else None
    if rep % 3 == 1:
        p = A.EinsumDotProver(A.MultilinearPolynomial.from_fr(L), A.MultilinearPolynomial.from_fr(R), None, A.EQ_NONE, 0, 0)
        t = A.Blake2bTranscript(b"x"); A.Sumcheck.prove(p, p.input_claim(), t, 18); p.free()
    t = A.Blake2bTranscript(b"relu_node")
    pr, cl, _ = NODE.prove_relu_node(tX, 16, r0, t)
    if t.state not in states_r and states_r:
        diff_report("relu", rep, pr, cl)
    first.setdefault("relu", (pr, cl))
    states_r[t.state] = states_r.get(t.state, 0) + 1
    t = A.Blake2bTranscript(b"einsum_node")
    pr, cl, _ = NODE.prove_einsum_node(tA, tB, 16, 1024, 4096, 14, r0, t)
    if t.state not in states_e and states_e:
        diff_report("einsum", rep, pr, cl)
    first.setdefault("einsum", (pr, cl))
    states_e[t.state] = states_e.get(t.state, 0) + 1
env = {k: os.environ.get(k) for k in ("ATLAS_LANE_EVENTS", "ATLAS_POOL_ANYSTREAM", "ATLAS_NO_POOL", "ATLAS_NO_LANE_STREAMS", "ATLAS_LANE_ONE_STREAM",
                                       "ATLAS_NO_MAIL_TAIL", "ATLAS_CH_HOST_POLL", "ATLAS_LANE_NO_GATE", "ATLAS_STRESS_PRE_STREAMS") if os.environ.get(k)}
ok = len(states_r) == 1 and len(states_e) == 1
print("stress_lanes", "OK" if ok else "NONDETERMINISTIC", env, "reps", reps, "relu states", sorted(states_r.values(), reverse=True), "einsum states", sorted(states_e.values(), reverse=True))
sys.exit(0 if ok else 1)
