"""Wall clock of the operator-level provers at 2^LOG_T elements (element-wise, selector-style, softmax)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import jolt_atlas_amd as A
from jolt_atlas_amd import instances as I
A.init(0)
n = int(os.environ.get("LOG_T", "20"))
ops = [A.MultilinearPolynomial.from_fr(A.random_fr(1 << n, 10 + i)) for i in range(6)]
r = A.random_fr(n, 1); claim = A.random_fr(1, 2)[0]
def T(make, reps=3):
    ts = []
    for _ in range(reps + 1):
        A.sync(); t0 = time.perf_counter(); inst = make(); t1 = time.perf_counter()
        inst.prove(claim, A.Blake2bTranscript(b"t")); t2 = time.perf_counter(); inst.free()
        ts.append((t1 - t0, t2 - t1))
    ts = ts[1:]
    return round(1e3 * float(np.median([a for a, _ in ts])), 3), round(1e3 * float(np.median([b for _, b in ts])), 3)
names = {0: "add", 3: "square", 4: "iff", 5: "mul", 6: "cube", 7: "div", 9: "dot(1 pair)", 10: "gather"}
nops = {0: 2, 3: 1, 4: 3, 5: 2, 6: 1, 7: 4, 9: 2, 10: 3}
for op, nm in names.items():
    k = A.random_fr(1, 3) if op == 10 else None
    print(f"elementwise {nm:12s} 2^{n}: new/prove ms", T(lambda: I.elementwise(op, ops[:nops[op]], r, k)))
K = 10
kq = A.MultilinearPolynomial.from_fr(A.random_fr(1 << K, 30))
print("softmax exp_sum   new/prove ms", T(lambda: I.softmax_instance(0, ops[0], None, K, n - K, A.random_fr(K, 4))))
print("softmax max       new/prove ms", T(lambda: I.softmax_instance(1, ops[0], ops[1], K, n - K, A.random_fr(K, 4))))
print("softmax recip     new/prove ms", T(lambda: I.softmax_instance(2, ops[0], kq, K, n - K, r)))
print("sum_axis          new/prove ms", T(lambda: I.softmax_instance(3, ops[0], None, 0, n, None)))
