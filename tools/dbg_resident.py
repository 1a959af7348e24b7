"""Stress of the resident bind-pass kernel against the launch-per-round passes: REPS proofs per size, every round polynomial and final claim
compared (the two are the same arithmetic: any difference is a visibility bug).  """
import os, sys, subprocess, json, hashlib
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
SIZES = [int(x) for x in os.environ.get("SIZES", "17,18,19,20,21,22").split(",")]
REPS = int(os.environ.get("REPS", "10"))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np
    import jolt_atlas_amd as A
    from oracle import orc
    A.init(0)
    out = {}
    for n in SIZES:
        hs = []
        for rep in range(REPS):
            L = orc.random_fr(1 << n, 100 + rep); R = orc.random_fr(1 << n, 200 + rep)
            prover = A.EinsumDotProver(A.MultilinearPolynomial.from_fr(L), A.MultilinearPolynomial.from_fr(R), None, A.EQ_NONE, 0, 0)
            t = A.Blake2bTranscript(b"dbg")
            claim = orc.random_fr(1, 5)[0]
            proof, ch, fin = A.Sumcheck.prove(prover, claim, t, n)
            prover.free()
            hs.append(hashlib.sha256(proof.tobytes() + fin.tobytes()).hexdigest()[:16])
        out[n] = hs
    print("RESULT " + json.dumps(out))
else:
    res = {}
    for env in ({"ATLAS_SC_RESIDENT": "1"}, {}):
        r = subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, **env), capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
        if not line: print(r.stderr[-2000:]); sys.exit(1)
        res["ATLAS_SC_RESIDENT" not in env] = json.loads(line[-1][7:])
    bad = 0
    for n in res[False]:
        a, b = res[False][n], res[True][n]
        nb = sum(1 for x, y in zip(a, b) if x != y)
        bad += nb
        print("n", n, "mismatching proofs", nb, "of", len(a))
    print("TOTAL_BAD", bad)
