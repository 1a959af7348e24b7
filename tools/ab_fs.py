"""A/B: whole-instance dot sumcheck with the transcript on the host (round channel) vs on the device."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import jolt_atlas_amd as A
A.init(0)
sizes = [int(x) for x in os.environ.get("SIZES", "6,10,12,14,16,18,20,22,24").split(",")]
for n in sizes:
    L = A.random_fr(1 << n, 1); R = A.random_fr(1 << n, 2)
    ml, mr = A.MultilinearPolynomial.from_fr(L), A.MultilinearPolynomial.from_fr(R)
    p = A.EinsumDotProver(ml.clone(), mr.clone()); claim = p.input_claim(); p.free()
    res = {}
    for mode, name in ((A.FS_HOST, "host"), (A.FS_DEVICE, "device")):
        A.set_fs_mode(mode)
        ts = []
        reps = 12 if n <= 22 else 5
        sets = [(ml.clone(), mr.clone()) for _ in range(reps)]
        out = None
        for i in range(reps):
            pr = A.EinsumDotProver(*sets[i]); t = A.Blake2bTranscript(b"ab")
            A.sync(); t0 = time.perf_counter()
            out = A.Sumcheck.prove(pr, claim, t, n)
            A.sync(); ts.append(time.perf_counter() - t0)
            pr.free()
        res[name] = (1e3 * float(np.median(ts[2:])), out, t.state)
        A.set_timing(True)
        pr = A.EinsumDotProver(ml.clone(), mr.clone()); A.Sumcheck.prove(pr, claim, A.Blake2bTranscript(b"ab"), n); pr.free()
        tm = A.last_timing(); A.set_timing(False)
        print(f"n={n} fs={name}: {res[name][0]:.4f} ms/instance ({1e3*res[name][0]/n:.2f} us/round); instrumented pass_ms {tm.pass_ms:.4f} fs_ms {tm.fs_ms:.4f} total {tm.total_ms:.4f} bytes {tm.pass_bytes}", flush=True)
    same = np.array_equal(res["host"][1][0], res["device"][1][0]) and res["host"][1][1] == res["device"][1][1] and res["host"][2] == res["device"][2]
    print(f"n={n} identical proofs: {same}", flush=True)
    ml.free(); mr.free()
A.set_fs_mode(A.FS_HOST)
