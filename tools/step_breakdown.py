"""Where a bench.py step spends its time: prover construction, prove, free — for both transcript placements."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import jolt_atlas_amd as A
A.init(0)
n = int(os.environ.get("N", "22"))
L = A.random_fr(1 << n, 1); R = A.random_fr(1 << n, 2)
ml, mr = A.MultilinearPolynomial.from_fr(L), A.MultilinearPolynomial.from_fr(R)
p = A.EinsumDotProver(ml.clone(), mr.clone()); claim = p.input_claim(); p.free()
for mode, name in ((A.FS_HOST, "host"), (A.FS_DEVICE, "device")):
    A.set_fs_mode(mode)
    K = 12
    sets = [(ml.clone(), mr.clone()) for _ in range(K)]
    A.sync()
    tc = tp = tf = 0.0
    t_all0 = time.perf_counter()
    for i in range(K):
        t0 = time.perf_counter()
        pr = A.EinsumDotProver(*sets[i]); t = A.Blake2bTranscript(b"ab")
        t1 = time.perf_counter()
        A.Sumcheck.prove(pr, claim, t, n)
        t2 = time.perf_counter()
        pr.free()
        t3 = time.perf_counter()
        if i >= 2: tc += t1 - t0; tp += t2 - t1; tf += t3 - t2
    A.sync()
    t_all = time.perf_counter() - t_all0
    print(f"fs={name}: construct {1e3*tc/(K-2):.4f} ms, prove {1e3*tp/(K-2):.4f} ms, free {1e3*tf/(K-2):.4f} ms; all {K} steps {1e3*t_all/K:.4f} ms/step", flush=True)
    # no free inside the loop
    sets = [(ml.clone(), mr.clone()) for _ in range(K)]
    provers = []
    A.sync(); t0 = time.perf_counter()
    for i in range(K):
        pr = A.EinsumDotProver(*sets[i]); t = A.Blake2bTranscript(b"ab")
        A.Sumcheck.prove(pr, claim, t, n); provers.append(pr)
    A.sync(); t1 = time.perf_counter()
    for pr in provers: pr.free()
    print(f"fs={name}: back-to-back proves without free {1e3*(t1-t0)/K:.4f} ms/step", flush=True)
