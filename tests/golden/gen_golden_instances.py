#!/usr/bin/env python3
"""Golden vectors of the operator-level sumcheck provers, from the dense-table Python models only (oracle/pymodel:
big-int arithmetic + hashlib BLAKE2b; neither the C oracle nor the device code is involved).

    python tests/golden/gen_golden_instances.py > tests/golden/instances.json

Each case stores its inputs (canonical integers, hex), the compressed round polynomials, the raw challenges and the
final transcript state.  tests/test_oracle_golden_instances.py replays them through the C oracle,
tests/test_gpu_golden_instances.py through the device."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from oracle.pymodel import field as F, ra as PR, sumcheck as PS          # noqa: E402
from oracle.pymodel.transcript import Blake2bTranscript                    # noqa: E402


def rand(n, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    return [int.from_bytes(rng.bytes(40), "little") % F.FR for _ in range(n)]


def small(n, seed, lo=-(1 << 15), hi=1 << 15):
    rng = np.random.default_rng(seed)
    return [int(v) % F.FR for v in rng.integers(lo, hi, size=n)]


def run(model, label):
    claim = model.input_claim()
    t = Blake2bTranscript(label)
    rows, rs, raw, last = PS.prove(model, claim, t)
    return dict(claim="%x" % claim, rows=[["%x" % c for c in r] for r in rows], challenges=["%x" % c for c in raw],
                state=t.state.hex(), finals=["%x" % c for c in model.finals()])


EW = {"add": (0, 2), "sub": (1, 2), "neg": (2, 1), "square": (3, 1), "iff": (4, 3), "mul": (5, 2), "cube": (6, 1), "div": (7, 4),
      "rsqrt": (8, 5), "dot": (9, 4), "gather": (10, 3), "hamming_bool": (11, 3), "teleport_div": (12, 3)}
cases = []
for name, (code, n_ops) in sorted(EW.items(), key=lambda kv: kv[1][0]):
    n_vars = 3
    ops = [small(1 << n_vars, 100 * code + i) for i in range(n_ops)]
    if name == "iff":
        ops[0] = [int(v) for v in np.random.default_rng(7).integers(0, 2, size=1 << n_vars)]
    consts = {"rsqrt": [(1 << 42) % F.FR, rand(1, 23)[0]], "gather": rand(1, 24), "hamming_bool": rand(3, 25), "teleport_div": [12345]}.get(name, [])
    r = rand(n_vars, 50 + code)
    out = run(PR.ElementwiseModel(name, ops, r, consts), b"golden_ew")
    cases.append(dict(family="elementwise", op=code, name=name, n_vars=n_vars, operands=[["%x" % v for v in o] for o in ops],
                      constants=["%x" % v for v in consts], r=["%x" % v for v in r], **out))
for kind, code, log_K, log_N in (("exp_sum", 0, 2, 2), ("max", 1, 1, 2), ("recip", 2, 2, 1), ("sum_axis", 3, 0, 3)):
    n = 1 << (log_K + log_N)
    a = small(n, 300 + code)
    b = None
    if kind == "max":
        b = [0] * n
        for k in range(1 << log_K):
            b[(k << log_N) + (k + 1) % (1 << log_N)] = 1
    if kind == "recip":
        b = small(1 << log_K, 310, 1, 1 << 20)
    r = rand(log_K + (log_N if kind == "recip" else 0), 320 + code) if kind != "sum_axis" else []
    out = run(PR.SoftmaxModel(kind, a, b, log_K, log_N, r), b"golden_sm")
    cases.append(dict(family="softmax", kind=code, name=kind, log_K=log_K, log_N=log_N, a=["%x" % v for v in a],
                      b=["%x" % v for v in b] if b is not None else None, r=["%x" % v for v in r], **out))
for N, shift, log_T in ((16, 3, 2), (32, 5, 1)):
    rng = np.random.default_rng(N + shift)
    idx = [int(x) for x in rng.integers(0, 1 << N, size=1 << log_T, dtype=np.uint64)]
    idx[0] = (1 << N) - 1
    r_node, gamma = rand(log_T, 3), rand(1, 4)[0] >> 130
    out = run(PR.PsRightShiftModel(idx, N, shift, r_node, gamma), b"golden_rs")
    cases.append(dict(family="ps_rshift", N=N, shift=shift, log_T=log_T, idx=["%x" % v for v in idx], r=["%x" % v for v in r_node],
                      gamma="%x" % gamma, **out))
json.dump(dict(generator="tests/golden/gen_golden_instances.py", cases=cases), sys.stdout, indent=0)
