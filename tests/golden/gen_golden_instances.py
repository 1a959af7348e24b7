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


def run(model, label, claim=None):
    claim = model.input_claim() if claim is None else claim
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
from oracle.pymodel import poly as P                                      # noqa: E402


def hx(v):
    return ["%x" % x for x in v]


# ---- one-hot "ra" family, openings, prefix-suffix lookups (models: RaVirtualModel ... PsUltModel)
rng = np.random.default_rng(77)
d, log_k, log_T = 3, 2, 3
H = [[int(v) for v in rng.integers(-1, 1 << log_k, size=1 << log_T)] for _ in range(d)]
chunks = [rand(log_k, 400 + i) for i in range(d)]
r_cycle = rand(log_T, 410)
cases.append(dict(family="ra_virtual", log_k=log_k, H=H, chunks=[hx(c) for c in chunks], r_cycle=hx(r_cycle),
                  **run(PR.RaVirtualModel(H, chunks, r_cycle), b"golden_ra")))
d, log_k, log_T = 2, 2, 2
H = [[int(v) for v in rng.integers(-1, 1 << log_k, size=1 << log_T)] for _ in range(d)]
gammas, r_address, r_cycle = [g >> 130 for g in rand(d, 420)], rand(log_k, 421), rand(log_T, 422)
E = P.eq_evals(r_cycle)
G = [[sum(E[j] for j in range(1 << log_T) if h[j] == k) % F.FR for k in range(1 << log_k)] for h in H]
cases.append(dict(family="booleanity", log_k=log_k, H=H, G=[hx(g) for g in G], gammas=hx(gammas), r_address=hx(r_address),
                  r_cycle=hx(r_cycle), **run(PR.BooleanityModel(H, log_k, gammas, r_address, r_cycle), b"golden_bool", 0)))
Gh, gp = [rand(1 << 2, 430 + i) for i in range(2)], rand(2, 432)
cases.append(dict(family="hamming", log_k=2, G=[hx(g) for g in Gh], gamma_powers=hx(gp), **run(PR.HammingModel(Gh, gp), b"golden_hw", sum(g * sum(row) for g, row in zip(gp, Gh)) % F.FR)))
poly, point = rand(1 << 4, 440), rand(4, 441)
_dm = PR.DenseOpeningModel(poly, point)
cases.append(dict(family="dense_opening", poly=hx(poly), point=hx(point),
                  **run(_dm, b"golden_do", sum(e * p_ for e, p_ in zip(_dm.eq, poly)) % F.FR)))
idx = [int(v) for v in rng.integers(-1, 4, size=8)]
ra_, rc_ = rand(2, 450), rand(3, 451)
cases.append(dict(family="onehot_opening", log_K=2, idx=idx, r_address=hx(ra_), r_cycle=hx(rc_),
                  **run(PR.OneHotOpeningModel(idx, 2, ra_, rc_), b"golden_oh")))
li = [int(v) for v in rng.integers(0, 1 << 16, size=4)]
li[0] = (1 << 16) - 1
r_node, gamma = rand(2, 460), rand(1, 461)[0] >> 130
cases.append(dict(family="ps_relu", N=16, idx=hx(li), r=hx(r_node), gamma="%x" % gamma, **run(PR.PsReluModel(li, 16, r_node, gamma), b"golden_relu")))
cases.append(dict(family="ps_clamp", N=16, bound=9, symmetric=1, idx=hx(li), r=hx(r_node), gamma="%x" % gamma,
                  **run(PR.PsClampModel(li, 16, 9, True, r_node, gamma), b"golden_clamp")))
l8 = [v & 0xff for v in li]
cases.append(dict(family="ps_identity", log_K=8, phases=4, idx=hx(l8), r=hx(r_node), **run(PR.PsIdentityModel(l8, 8, r_node), b"golden_id")))
lu = [int(v) for v in rng.integers(0, 1 << 63, size=2)]
r1 = rand(1, 470)
cases.append(dict(family="ps_ult", idx=hx(lu), r=hx(r1), gamma="%x" % gamma, **run(PR.PsUltModel(lu, r1, gamma), b"golden_ult")))
json.dump(dict(generator="tests/golden/gen_golden_instances.py", cases=cases), sys.stdout, indent=0)
