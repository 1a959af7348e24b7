"""CPU: oracle/ra.c (Gruen split-eq, product grid + Toom interpolation, expanding table) against
the naive full-table Python models in oracle/pymodel/ra.py."""
import numpy as np
import pytest

from oracle import orc, orc_ra as OR
from oracle.pymodel import field as F, ra as PR, sumcheck as PS
from oracle.pymodel.transcript import Blake2bTranscript


def _rand(n, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    return [int.from_bytes(rng.bytes(40), "little") % F.FR for _ in range(n)]


def _indices(d, T, K, seed, none_frac=0.15):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(d):
        a = rng.integers(0, K, size=T).astype(np.int32)
        a[rng.random(T) < none_frac] = -1
        out.append(a)
    return out


def _prove_py(model, claim, label):
    t = Blake2bTranscript(label)
    rows, rs, raw, last = PS.prove(model, claim, t)
    return rows, raw, t


@pytest.mark.parametrize("d,log_k,log_T", [(1, 2, 3), (2, 4, 4), (3, 4, 5), (4, 2, 1), (8, 4, 3), (16, 4, 2), (5, 3, 6)])
def test_ra_virtual_oracle_matches_naive_model(d, log_k, log_T):
    T, K = 1 << log_T, 1 << log_k
    H = _indices(d, T, K, 7 * d + log_T)
    chunks = [_rand(log_k, 100 + i) for i in range(d)]
    r_cycle = _rand(log_T, 55)
    model = PR.RaVirtualModel([list(map(int, h)) for h in H], chunks, r_cycle)
    claim = model.input_claim()
    rows_p, raw_p, tp = _prove_py(model, claim, b"ra_virtual")
    inst = OR.ra_virtual(H, log_k, np.stack([orc.from_ints(c) for c in chunks]), orc.from_ints(r_cycle))
    to = orc.new_transcript(b"ra_virtual")
    rows_o, raw_o = inst.prove(orc.from_ints([claim])[0], to)
    assert raw_o == raw_p
    assert [orc.to_ints(r) for r in rows_o] == rows_p
    assert bytes(to.state) == tp.state


@pytest.mark.parametrize("d,log_k,log_T", [(1, 1, 1), (2, 4, 3), (3, 2, 4), (4, 4, 2), (8, 4, 1)])
def test_booleanity_oracle_matches_naive_model(d, log_k, log_T):
    T, K = 1 << log_T, 1 << log_k
    H = _indices(d, T, K, 11 * d + log_T)
    gammas = [g >> 130 for g in _rand(d, 3)]          # challenge-sized values
    r_address, r_cycle = _rand(log_k, 4), _rand(log_T, 5)
    model = PR.BooleanityModel([list(map(int, h)) for h in H], log_k, gammas, r_address, r_cycle)
    rows_p, raw_p, tp = _prove_py(model, 0, b"booleanity")
    G = OR.ra_G(H, log_k, orc.from_ints(r_cycle))
    inst = OR.booleanity(G, H, log_k, orc.from_ints(gammas), orc.from_ints(r_address), orc.from_ints(r_cycle))
    to = orc.new_transcript(b"booleanity")
    rows_o, raw_o = inst.prove(orc.from_ints([0])[0], to)
    assert raw_o == raw_p
    assert [orc.to_ints(r) for r in rows_o] == rows_p
    assert bytes(to.state) == tp.state


@pytest.mark.parametrize("d,log_k", [(1, 1), (2, 4), (8, 4), (3, 8)])
def test_hamming_oracle_matches_naive_model(d, log_k):
    K = 1 << log_k
    G = [_rand(K, 20 + i) for i in range(d)]
    gp = _rand(d, 9)
    claim = sum(g * sum(row) for g, row in zip(gp, G)) % F.FR
    model = PR.HammingModel(G, gp)
    rows_p, raw_p, tp = _prove_py(model, claim, b"hamming")
    inst = OR.hamming(np.stack([orc.from_ints(g) for g in G]), log_k, orc.from_ints(gp))
    to = orc.new_transcript(b"hamming")
    rows_o, raw_o = inst.prove(orc.from_ints([claim])[0], to)
    assert raw_o == raw_p
    assert [orc.to_ints(r) for r in rows_o] == rows_p
    assert bytes(to.state) == tp.state


def test_degenerate_all_none_indices_trim_to_zero_polys():
    d, log_k, log_T = 2, 2, 2
    H = [np.full(4, -1, dtype=np.int32) for _ in range(d)]
    chunks = [_rand(log_k, 1), _rand(log_k, 2)]
    r_cycle = _rand(log_T, 3)
    model = PR.RaVirtualModel([list(map(int, h)) for h in H], chunks, r_cycle)
    rows_p, raw_p, tp = _prove_py(model, 0, b"zero")
    inst = OR.ra_virtual(H, log_k, np.stack([orc.from_ints(c) for c in chunks]), orc.from_ints(r_cycle))
    to = orc.new_transcript(b"zero")
    rows_o, raw_o = inst.prove(orc.from_ints([0])[0], to)
    assert [orc.to_ints(r) for r in rows_o] == rows_p and raw_o == raw_p
    assert all(r == [0] for r in rows_p)


def test_batched_mix_of_ra_instances_matches_python_model():
    """BatchedSumcheck over booleanity + hamming weight + ra virtualisation + a dot instance, the
    grouping the lookup ops use (sumcheck.rs:30-184)."""
    from oracle import orc_batched as OB
    from oracle.pymodel import batched as PB
    d, log_k, log_T = 2, 2, 3
    T, K = 1 << log_T, 1 << log_k
    H = _indices(d, T, K, 99)
    Hl = [list(map(int, h)) for h in H]
    chunks = [_rand(log_k, 1), _rand(log_k, 2)]
    r_cycle, r_address = _rand(log_T, 3), _rand(log_k, 4)
    gammas, gp = [g >> 130 for g in _rand(d, 5)], _rand(d, 6)
    Gm = [[0] * K for _ in range(d)]
    from oracle.pymodel import poly as P
    E = P.eq_evals(r_cycle)
    for i in range(d):
        for j, k in enumerate(Hl[i]):
            if k >= 0:
                Gm[i][k] = (Gm[i][k] + E[j]) % F.FR
    L, R = _rand(16, 7), _rand(16, 8)
    m_ra = PR.RaVirtualModel(Hl, chunks, r_cycle)
    models = [PR.BooleanityModel(Hl, log_k, gammas, r_address, r_cycle), PR.HammingModel(Gm, gp), m_ra, PS.DotProver(L, R)]
    claims = [0, sum(g * sum(row) for g, row in zip(gp, Gm)) % F.FR, m_ra.input_claim(), sum(a * b for a, b in zip(L, R)) % F.FR]
    tp = Blake2bTranscript(b"mix")
    rows_p, raw_p, _ = PB.prove(PB.prepare(models), claims, tp)
    G = np.stack([orc.from_ints(g) for g in Gm])
    o = [OB.ra_instance(OR.booleanity(G, H, log_k, orc.from_ints(gammas), orc.from_ints(r_address), orc.from_ints(r_cycle)), orc.from_ints([claims[0]])[0]),
         OB.ra_instance(OR.hamming(G, log_k, orc.from_ints(gp)), orc.from_ints([claims[1]])[0]),
         OB.ra_instance(OR.ra_virtual(H, log_k, np.stack([orc.from_ints(c) for c in chunks]), orc.from_ints(r_cycle)), orc.from_ints([claims[2]])[0]),
         OB.dot_instance(orc.from_ints(L), orc.from_ints(R), orc.from_ints([claims[3]])[0])]
    to = orc.new_transcript(b"mix")
    rows_o, raw_o, _ = OB.batched_prove(o, to)
    assert raw_o == raw_p
    assert [orc.to_ints(r) for r in rows_o] == rows_p
    assert bytes(to.state) == tp.state


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 7])
def test_dense_opening_oracle_matches_naive_model(n):
    poly, point = _rand(1 << n, 31 + n), _rand(n, 32 + n)
    model = PR.DenseOpeningModel(poly, point)
    claim = sum(e * p for e, p in zip(model.eq, poly)) % F.FR
    rows_p, raw_p, tp = _prove_py(model, claim, b"dense_opening")
    inst = OR.dense_opening(orc.from_ints(poly), orc.from_ints(point))
    to = orc.new_transcript(b"dense_opening")
    rows_o, raw_o = inst.prove(orc.from_ints([claim])[0], to)
    assert raw_o == raw_p
    assert [orc.to_ints(r) for r in rows_o] == rows_p
    assert bytes(to.state) == tp.state


@pytest.mark.parametrize("log_K,log_T", [(1, 1), (2, 3), (4, 2), (4, 5), (3, 4), (8, 1)])
def test_onehot_opening_oracle_matches_naive_model(log_K, log_T):
    idx = _indices(1, 1 << log_T, 1 << log_K, 5 * log_K + log_T)[0]
    r_address, r_cycle = _rand(log_K, 41), _rand(log_T, 42)
    model = PR.OneHotOpeningModel(list(map(int, idx)), log_K, r_address, r_cycle)
    claim = model.input_claim()
    rows_p, raw_p, tp = _prove_py(model, claim, b"onehot_opening")
    inst = OR.onehot_opening(idx, log_K, orc.from_ints(r_address), orc.from_ints(r_cycle))
    to = orc.new_transcript(b"onehot_opening")
    rows_o, raw_o = inst.prove(orc.from_ints([claim])[0], to)
    assert raw_o == raw_p
    assert [orc.to_ints(r) for r in rows_o] == rows_p
    assert bytes(to.state) == tp.state


@pytest.mark.parametrize("n,N", [(1, 2), (3, 2), (4, 3), (3, 4), (2, 5), (5, 3)])
def test_eval_reduction_oracle_properties(n, N):
    """h(i) = claim_i, h(t) = P(l(t)) at a random t, degree bound, and the reduced claim P(r') = v'
    (evaluation_reduction.rs:129-137, 186-204), all checked with the plain-int model."""
    from oracle.pymodel import poly as P
    mle = _rand(1 << n, 61 + n)
    pts = [_rand(n, 70 + j) for j in range(N)]
    claims = [P.evaluate(mle, p) for p in pts]
    to = orc.new_transcript(b"eval_reduction")
    h, r, c = OR.eval_reduction_prove(orc.from_ints(mle), np.stack([orc.from_ints(p) for p in pts]), orc.from_ints(claims), to)
    hc = orc.to_ints(h)
    assert len(hc) <= n * (N - 1) + 1 and len(hc) == n * (N - 1) + 1      # generic inputs reach the bound
    ev = lambda t: sum(cf * pow(t, k, F.FR) for k, cf in enumerate(hc)) % F.FR
    for i in range(N):
        assert ev(i) == claims[i]
    # l(t): per-variable interpolation through (j, pts[j][i])
    tt = _rand(1, 5)[0]
    lt = [sum(cf * pow(tt, k, F.FR) for k, cf in enumerate(PR.interpolate([pts[j][i] for j in range(N)]))) % F.FR for i in range(n)]
    assert ev(tt) == P.evaluate(mle, lt)
    # transcript + reduced instance
    tp = Blake2bTranscript(b"eval_reduction")
    tp.append_message(b"UncompressedUniPoly_begin")
    for cf in hc:
        tp.append_scalar(cf)
    tp.append_message(b"UncompressedUniPoly_end")
    x = F.challenge_to_fr(tp.challenge_u128())
    assert bytes(to.state) == tp.state
    rr = orc.to_ints(r)
    assert rr == [sum(cf * pow(x, k, F.FR) for k, cf in enumerate(PR.interpolate([pts[j][i] for j in range(N)]))) % F.FR for i in range(n)]
    assert orc.to_ints(c)[0] == ev(x) == P.evaluate(mle, rr)


def test_eval_reduction_structured_inputs_trim():
    """zero-padded coefficients and points sharing coordinates: h is the trimmed exact polynomial."""
    from oracle.pymodel import poly as P
    n, N = 3, 3
    mle = _rand(3, 1) + [0] * 5
    base = _rand(n, 2)
    pts = [list(base), [base[0], _rand(1, 3)[0], base[2]], [base[0], _rand(1, 4)[0], base[2]]]
    claims = [P.evaluate(mle, p) for p in pts]
    to = orc.new_transcript(b"eval_reduction")
    h, r, c = OR.eval_reduction_prove(orc.from_ints(mle), np.stack([orc.from_ints(p) for p in pts]), orc.from_ints(claims), to)
    hc = orc.to_ints(h)
    assert hc[-1] != 0 or hc == [0]
    assert len(hc) <= 1 * (N - 1) + 1          # only variable 1 moves along l
    assert P.evaluate(mle, orc.to_ints(r)) == orc.to_ints(c)[0]
    # single opening: short path, no transcript interaction
    t1 = orc.new_transcript(b"x")
    h1, r1, c1 = OR.eval_reduction_prove(orc.from_ints(mle), np.stack([orc.from_ints(pts[0])]), orc.from_ints(claims[:1]), t1)
    assert orc.to_ints(h1) == [claims[0]] and orc.to_ints(r1) == pts[0] and t1.n_rounds == 0


@pytest.mark.parametrize("N,log_T", [(16, 1), (16, 3), (32, 2), (32, 4)])
def test_ps_shout_relu_oracle_matches_closed_form_model(N, log_T):
    """oracle/psshout.c (prefix checkpoints, suffix Q tables, phases, expanding tables, Gruen) against
    the closed-form model; also pins ReLU's table MLE on boolean points."""
    T = 1 << log_T
    rng = np.random.default_rng(N + log_T)
    idx = [int(x) for x in rng.integers(0, 1 << N, size=T, dtype=np.uint64)]
    idx[0] = (1 << N) - 1                      # negative value, all ones
    if T > 2:
        idx[1] = 0; idx[2] = 1 << (N - 1)      # zero; most negative
    r_node, gamma = _rand(log_T, 3), _rand(1, 4)[0] >> 130
    model = PR.PsReluModel(idx, N, r_node, gamma)
    # ReluTable::materialize_entry vs the MLE on boolean inputs
    for k in idx:
        bits = [(k >> (N - 1 - i)) & 1 for i in range(N)]
        signed = k - (1 << N) if k >> (N - 1) else k
        assert model._W(bits) == (max(0, signed) + gamma * signed) % F.FR
    claim = model.input_claim()
    rows_p, raw_p, tp = _prove_py(model, claim, b"ps_relu")
    inst = OR.ps_relu(idx, N, orc.from_ints(r_node), orc.from_ints([gamma])[0])
    to = orc.new_transcript(b"ps_relu")
    rows_o, raw_o = inst.prove(orc.from_ints([claim])[0], to)
    assert raw_o == raw_p
    assert [orc.to_ints(r) for r in rows_o] == rows_p
    assert bytes(to.state) == tp.state


@pytest.mark.parametrize("N,shift,log_T", [(16, 1, 2), (16, 5, 3), (32, 1, 2), (32, 3, 3), (32, 7, 1), (32, 0, 2), (16, 15, 2)])
def test_ps_shout_right_shift_oracle_matches_closed_form_model(N, shift, log_T):
    """RightShiftTable<N> (k >> D, lookup_tables/right_shift.rs) through the same unary read-raf prover: the
    oracle's literal prefix checkpoints / suffix tables against the closed-form model."""
    T = 1 << log_T
    rng = np.random.default_rng(N + shift + log_T)
    idx = [int(x) for x in rng.integers(0, 1 << N, size=T, dtype=np.uint64)]
    idx[0] = (1 << N) - 1
    if T > 2:
        idx[1] = 0; idx[2] = (1 << shift) - 1 if shift else 1      # everything shifted out
    r_node, gamma = _rand(log_T, 3), _rand(1, 4)[0] >> 130
    model = PR.PsRightShiftModel(idx, N, shift, r_node, gamma)
    for k in idx:                               # materialize_entry vs the MLE on boolean inputs
        bits = [(k >> (N - 1 - i)) & 1 for i in range(N)]
        signed = k - (1 << N) if k >> (N - 1) else k
        assert model._W(bits) == ((k >> shift) + gamma * signed) % F.FR
    claim = model.input_claim()
    rows_p, raw_p, tp = _prove_py(model, claim, b"ps_rshift")
    inst = OR.ps_rshift(idx, N, shift, orc.from_ints(r_node), orc.from_ints([gamma])[0])
    to = orc.new_transcript(b"ps_rshift")
    rows_o, raw_o = inst.prove(orc.from_ints([claim])[0], to)
    assert raw_o == raw_p
    assert [orc.to_ints(r) for r in rows_o] == rows_p
    assert bytes(to.state) == tp.state


@pytest.mark.parametrize("log_K,phases,log_T", [(8, 4, 2), (16, 8, 3), (16, 2, 1), (12, 3, 4), (32, 8, 2)])
def test_identity_range_check_oracle_matches_closed_form_model(log_K, phases, log_T):
    T = 1 << log_T
    rng = np.random.default_rng(log_K + phases + log_T)
    idx = [int(x) for x in rng.integers(0, 1 << log_K, size=T, dtype=np.uint64)]
    idx[0] = (1 << log_K) - 1
    r_node = _rand(log_T, 3)
    model = PR.PsIdentityModel(idx, log_K, r_node)
    claim = model.input_claim()
    rows_p, raw_p, tp = _prove_py(model, claim, b"identity_rc")
    inst = OR.ps_identity(idx, log_K, phases, orc.from_ints(r_node))
    to = orc.new_transcript(b"identity_rc")
    rows_o, raw_o = inst.prove(orc.from_ints([claim])[0], to)
    assert raw_o == raw_p
    assert [orc.to_ints(r) for r in rows_o] == rows_p
    assert bytes(to.state) == tp.state


@pytest.mark.parametrize("N,bound,sym,log_T", [(16, 9, True, 2), (16, 5, False, 3), (32, 9, True, 3), (32, 17, True, 2), (32, 12, False, 2), (64, 31, True, 2)])
def test_ps_shout_clamp_oracle_matches_closed_form_model(N, bound, sym, log_T):
    T = 1 << log_T
    rng = np.random.default_rng(N + bound + log_T)
    vals = [int(v) for v in rng.integers(-(1 << (bound + 2)), 1 << (bound + 2), size=T)]
    vals[0] = -1; vals[1] = (1 << bound) - 1
    if T > 2:
        vals[2] = -(1 << bound); vals[3] = -(1 << (N - 1))
    idx = [v & ((1 << N) - 1) for v in vals]
    r_node, gamma = _rand(log_T, 3), _rand(1, 4)[0] >> 130
    model = PR.PsClampModel(idx, N, bound, sym, r_node, gamma)
    # materialize_entry (clamp.rs:124-136) vs the MLE on boolean points
    for v, k in zip(vals, idx):
        bits = [(k >> (N - 1 - i)) & 1 for i in range(N)]
        lo = -(1 << bound) if sym else 0
        want = min(max(v, lo), (1 << bound) - 1)
        assert model._W(bits) == (want + gamma * v) % F.FR
    claim = model.input_claim()
    rows_p, raw_p, tp = _prove_py(model, claim, b"ps_clamp")
    inst = OR.ps_clamp(idx, N, bound, sym, orc.from_ints(r_node), orc.from_ints([gamma])[0])
    to = orc.new_transcript(b"ps_clamp")
    rows_o, raw_o = inst.prove(orc.from_ints([claim])[0], to)
    assert raw_o == raw_p
    assert [orc.to_ints(r) for r in rows_o] == rows_p
    assert bytes(to.state) == tp.state


@pytest.mark.parametrize("log_T", [1, 3])
def test_ps_shout_unsigned_less_than_oracle_matches_closed_form_model(log_T):
    """binary read-raf (ps_shout/binary.rs) with UnsignedLessThanTable<32>: 64 interleaved address bits."""
    T = 1 << log_T
    rng = np.random.default_rng(log_T)
    xs = [int(v) for v in rng.integers(0, 1 << 32, size=T, dtype=np.uint64)]
    ys = [int(v) for v in rng.integers(0, 1 << 32, size=T, dtype=np.uint64)]
    xs[0], ys[0] = 5, 5                               # equal operands
    xs[1], ys[1] = (1 << 32) - 1, 0
    if T > 2:
        xs[2], ys[2] = 7, (1 << 31); xs[3], ys[3] = ys[3] & ~0xff, (ys[3] & ~0xff) + 1     # differ only in the low byte
        xs[3] = ys[3] - 1
    idx = [OR.interleave(x, y) for x, y in zip(xs, ys)]
    r_node, gamma = _rand(log_T, 3), _rand(1, 4)[0] >> 130
    model = PR.PsUltModel(idx, r_node, gamma)
    sgn = lambda v: v - (1 << 32) if v >> 31 else v
    for k, x, y in zip(idx, xs, ys):
        bits = [(k >> (63 - i)) & 1 for i in range(64)]
        assert model._W(bits) == (int(x < y) + gamma * sgn(x) + gamma * gamma * sgn(y)) % F.FR
    claim = model.input_claim()
    rows_p, raw_p, tp = _prove_py(model, claim, b"ps_ult")
    inst = OR.ps_ult(idx, orc.from_ints(r_node), orc.from_ints([gamma])[0])
    to = orc.new_transcript(b"ps_ult")
    rows_o, raw_o = inst.prove(orc.from_ints([claim])[0], to)
    assert raw_o == raw_p
    assert [orc.to_ints(r) for r in rows_o] == rows_p
    assert bytes(to.state) == tp.state


_EW = {"add": (0, 2), "sub": (1, 2), "neg": (2, 1), "square": (3, 1), "iff": (4, 3), "mul": (5, 2), "cube": (6, 1),
       "div": (7, 4), "rsqrt": (8, 5), "dot": (9, 6), "gather": (10, 3), "hamming_bool": (11, 5), "teleport_div": (12, 3)}


@pytest.mark.parametrize("op", sorted(_EW))
@pytest.mark.parametrize("n_vars", [1, 2, 5])
def test_elementwise_oracle_matches_naive_model(op, n_vars):
    """oracle/elementwise.c (Gruen split-eq fold, gruen_poly_deg_2/3, product-sum finish for Cube) against the
    dense-table model, transcript included."""
    code, n_ops = _EW[op]
    n = 1 << n_vars
    rng = np.random.default_rng(code * 10 + n_vars)
    # quantised tensors: small signed integers as field elements; the Iff mask is 0/1
    operands = [[int(v) % F.FR for v in rng.integers(-(1 << 15), 1 << 15, size=n)] for _ in range(n_ops)]
    if op == "iff":
        operands[0] = [int(v) for v in rng.integers(0, 2, size=n)]
    r_node = _rand(n_vars, 17)
    consts = [(1 << 42) % F.FR, _rand(1, 23)[0]] if op == "rsqrt" else [_rand(1, 23)[0]] if op == "gather" else _rand(5, 29) if op == "hamming_bool" else [12345] if op == "teleport_div" else []
    model = PR.ElementwiseModel(op, operands, r_node, consts)
    claim = model.input_claim()
    rows_p, raw_p, tp = _prove_py(model, claim, b"elementwise")
    inst = OR.elementwise(code, [orc.from_ints(o) for o in operands], orc.from_ints(r_node), orc.from_ints(consts) if consts else None)
    to = orc.new_transcript(b"elementwise")
    rows_o, raw_o = inst.prove(orc.from_ints([claim])[0], to)
    assert raw_o == raw_p
    assert [orc.to_ints(r) for r in rows_o] == rows_p
    assert bytes(to.state) == tp.state
    assert orc.to_ints(inst.finals()) == model.finals()


@pytest.mark.parametrize("kind,log_K,log_N", [("exp_sum", 2, 3), ("exp_sum", 1, 1), ("exp_sum", 3, 2), ("max", 2, 3), ("max", 1, 2),
                                              ("recip", 2, 3), ("recip", 3, 1), ("recip", 1, 4), ("sum_axis", 0, 4), ("sum_axis", 0, 1)])
def test_softmax_and_sum_axis_oracle_matches_dense_model(kind, log_K, log_N):
    """oracle/softmax.c (K-entry eq table in phase 1, Gruen split-eq in phase 2, degree-1 hint messages) against the
    dense-table model."""
    code = {"exp_sum": 0, "max": 1, "recip": 2, "sum_axis": 3}[kind]
    n = 1 << (log_K + log_N)
    rng = np.random.default_rng(code * 100 + log_K * 10 + log_N)
    a = [int(v) % F.FR for v in rng.integers(-(1 << 15), 1 << 15, size=n)]
    b = None
    if kind == "max":                          # one-hot argmax indicator per row
        b = [0] * n
        for k in range(1 << log_K):
            b[(k << log_N) + int(rng.integers(0, 1 << log_N))] = 1
    if kind == "recip":
        b = [int(v) for v in rng.integers(1, 1 << 20, size=1 << log_K)]
    r = _rand(log_K + (log_N if kind == "recip" else 0), 31) if kind != "sum_axis" else []
    model = PR.SoftmaxModel(kind, a, b, log_K, log_N, r)
    claim = model.input_claim()
    rows_p, raw_p, tp = _prove_py(model, claim, b"softmax")
    inst = OR.softmax(code, orc.from_ints(a), orc.from_ints(b) if b is not None else None, log_K, log_N,
                      orc.from_ints(r) if r else None)
    to = orc.new_transcript(b"softmax")
    rows_o, raw_o = inst.prove(orc.from_ints([claim])[0], to)
    assert raw_o == raw_p
    assert [orc.to_ints(x) for x in rows_o] == rows_p
    assert bytes(to.state) == tp.state
    assert orc.to_ints(inst.finals()) == model.finals()


@pytest.mark.parametrize("log_retained,log_reduced", [(2, 3), (0, 4), (3, 1), (1, 1)])
def test_mean_of_squares_is_the_dot_prover_with_eq_high(log_retained, log_reduced):
    """MeanOfSquaresReductionProver (ops/mean_of_squares.rs:363-398) = EinsumDotProver with left = right = operand and
    EqSchedule::High { log_eq = log_retained, low_bits = log_reduced }: the oracle's dot prover against the dense model."""
    n = 1 << (log_retained + log_reduced)
    rng = np.random.default_rng(10 * log_retained + log_reduced)
    a = [int(v) % F.FR for v in rng.integers(-(1 << 15), 1 << 15, size=n)]
    r = _rand(log_retained, 41)
    model = PR.SoftmaxModel("mean_sq", a, None, log_retained, log_reduced, r)
    claim = model.input_claim()
    rows_p, raw_p, tp = _prove_py(model, claim, b"mean_sq")
    L = orc.from_ints(a)
    eq = orc.eq_evals(orc.from_ints(r)) if log_retained else orc.from_ints([1])
    to = orc.new_transcript(b"mean_sq")
    proof, ch, fin = orc.sumcheck_dot_prove(L, L.copy(), orc.from_ints([claim]), to, eq, 1, log_retained, log_reduced)
    assert ch == raw_p
    assert [orc.to_ints(row) for row in proof] == rows_p
    assert bytes(to.state_bytes()) == tp.state
    assert orc.to_ints(fin[:1]) == model.finals()[:1]


def test_ra_claim_helper_matches_bit_products():
    """orc_ra_claim (the C shortcut the bench-size node tests use) against the definition: sum_j eq(r_cycle, j) prod_i (bit_i ? r_i : 1 - r_i)"""
    from oracle import graph as OG
    rng = np.random.default_rng(9)
    for log_K, log_T in ((14, 5), (32, 3), (64, 4), (9, 6)):
        idx = rng.integers(0, 1 << min(log_K, 62), size=1 << log_T).astype(np.uint64)
        if log_K == 64:
            idx |= np.uint64(1) << np.uint64(63)
        pt = orc.random_fr(log_K + log_T, 40 + log_K)
        want = orc.evaluate(np.stack([OG.eq_bits(pt[:log_K], v, log_K) for v in idx]), np.ascontiguousarray(pt[log_K:]))
        assert np.array_equal(OR.ra_claim(idx, log_K, pt), want)


def test_shout_batch_member_matches_standalone_prover():
    """the dense Shout ReadRafProver as an instance (orc_shout_inst_*, a member of the softmax stage-3 batch) against the reference's
    three-polynomial loop run on its own (orc_sumcheck_readraf_prove)"""
    import ctypes as C
    rng = np.random.default_rng(11)
    log_K, log_T = 5, 6
    K, T = 1 << log_K, 1 << log_T
    idx = rng.integers(0, K, size=T).astype(np.uint64)
    table = rng.integers(-1000, 1000, size=K).astype(np.int32)
    r = orc.random_fr(log_T, 3); gamma = orc.random_fr(1, 4)[0]
    E = orc.eq_evals(r)
    G = orc.fr_array(K)
    orc.lib.orc_shout_G(idx.ctypes.data_as(C.c_void_p), C.c_size_t(T), C.c_size_t(log_K), orc._p(E), orc._p(G))
    claim = orc.fr_array(1)
    orc.lib.orc_readraf_claim(orc._p(G), table.ctypes.data_as(C.c_void_p), C.c_size_t(log_K), orc._p(gamma.reshape(1, 4)), orc._p(claim))
    t1 = orc.new_transcript(b"shout"); t2 = orc.new_transcript(b"shout")
    proof = orc.fr_array(2 * log_K); ch = np.zeros(2 * log_K, dtype=np.uint64); fin = orc.fr_array(1)
    orc.lib.orc_sumcheck_readraf_prove(orc._p(G.copy()), table.ctypes.data_as(C.c_void_p), C.c_size_t(log_K), orc._p(gamma.reshape(1, 4)), orc._p(claim),
                                       C.byref(t1), orc._p(proof), orc._p(ch), orc._p(fin))
    I = OR.shout_read_raf(idx, table, log_K, r, gamma)
    rows, ch2 = I.prove(claim[0], t2)
    assert np.array_equal(np.concatenate(rows), proof)
    assert t1.state_bytes() == t2.state_bytes()
    assert np.array_equal(I.final(), fin[0])
