import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import jolt_atlas_amd as A
A.init(0)
for n in (int(x) for x in os.environ.get("SIZES", "10,22").split(",")):
    L = A.random_fr(1 << n, 1); R = A.random_fr(1 << n, 2)
    ml, mr = A.MultilinearPolynomial.from_fr(L), A.MultilinearPolynomial.from_fr(R)
    p = A.EinsumDotProver(ml.clone(), mr.clone()); claim = p.input_claim(); p.free()
    for rep in range(3):
        pr = A.EinsumDotProver(ml.clone(), mr.clone()); t = A.Blake2bTranscript(b"ab")
        A.sync()
        A.Sumcheck.prove(pr, claim, t, n); pr.free()
