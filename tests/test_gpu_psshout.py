"""GPU parity: prefix-suffix Shout read-raf sumcheck (unary, ReLU table) vs oracle/psshout.c."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["pipelined", "hoststepped"])
def stepping(request, monkeypatch):
    """Every test runs twice: over the round channel (all launches enqueued up front, transcript on the host thread) and
    host-stepped through compute_message / ingest_challenge (ATLAS_NO_PIPELINE=1)."""
    if request.param == "hoststepped":
        monkeypatch.setenv("ATLAS_NO_PIPELINE", "1")
    else:
        monkeypatch.delenv("ATLAS_NO_PIPELINE", raising=False)
    return request.param


def _claim(orc, idx, N, r_node, gamma):
    """rv_claim + gamma * operand_claim = sum_t eq(r_node, t) (relu(x_t) + gamma x_t), x_t signed."""
    E = orc.eq_evals(r_node)
    acc = orc.fr_array(1)[0]
    for t, k in enumerate(idx):
        k = int(k)
        x = k - (1 << N) if k >> (N - 1) else k
        w = orc.fr_array(1)
        orc.lib.fr_from_i64(x, orc._p(w))
        rv = orc.from_ints([max(0, x)])[0]
        term = orc.fr_add_arr(rv, orc.fr_mul_arr(gamma, w[0]))
        acc = orc.fr_add_arr(acc, orc.fr_mul_arr(E[t], term))
    return acc


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("N,log_T", [(16, 1), (16, 6), (32, 3), (32, 10), (32, 13)])
def test_ps_shout_relu_bit_exact(atlas, N, log_T, mode):
    import ctypes as C
    from oracle import orc, orc_ra as OR
    from jolt_atlas_amd import instances as I
    A = atlas
    orc.lib.fr_from_i64.argtypes = [C.c_int64, C.c_void_p]
    A.set_challenge_mode(mode); orc.lib.orc_set_challenge_mode(mode)
    try:
        T = 1 << log_T
        rng = np.random.default_rng(N * 100 + log_T)
        # activations: mostly small magnitudes of both signs, plus extremes
        small = rng.integers(-(1 << 14), 1 << 14, size=T, dtype=np.int64)
        idx = (small & ((1 << N) - 1)).astype(np.uint64)
        idx[0] = (1 << N) - 1
        if T > 2:
            idx[1] = 0; idx[2] = 1 << (N - 1)
        if T > 4:
            idx[3] = (1 << (N - 1)) - 1
        r_node, gamma = orc.random_fr(log_T, 5), orc.random_fr(1, 6)[0]
        claim = _claim(orc, idx, N, r_node, gamma)
        t_o = orc.new_transcript(b"ps_relu")
        rows_o, ch_o = OR.ps_relu(idx, N, r_node, gamma).prove(claim, t_o)
        inst = I.ps_shout_relu(idx, N, r_node, gamma)
        assert inst.num_rounds() == N + log_T and inst.degree() == 2
        t_g = A.Blake2bTranscript(b"ps_relu")
        rows_g, ch_g = inst.prove(claim, t_g)
        assert ch_g == ch_o
        assert len(rows_g) == len(rows_o) and all(np.array_equal(a, b) for a, b in zip(rows_g, rows_o))
        assert t_g.state == t_o.state_bytes()
        # final claim: ra(r_address, r_cycle) — MLE over t of eq(r_address, idx_t) at the reversed cycle challenges
        rs = orc.challenges_to_fr(ch_g)
        if log_T <= 6:
            r_addr, r_cyc = rs[:N], rs[N:][::-1]
            vals = []
            for k in idx:
                w = orc.from_ints([1])[0]
                for i in range(N):
                    bit = (int(k) >> (N - 1 - i)) & 1
                    f = r_addr[i] if bit else orc.fr_add_arr(orc.from_ints([1])[0], orc.fr_mul_arr(orc.from_ints([F_MINUS_ONE])[0], r_addr[i]))
                    w = orc.fr_mul_arr(w, f)
                vals.append(w)
            assert np.array_equal(inst.final_claims()[0], orc.evaluate(np.stack(vals), np.ascontiguousarray(r_cyc)))
        inst.free()
    finally:
        A.set_challenge_mode(0); orc.lib.orc_set_challenge_mode(0)


F_MINUS_ONE = 21888242871839275222246405745257275088548364400416034343698204186575808495616


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("N,shift,log_T", [(16, 1, 1), (16, 5, 6), (32, 1, 3), (32, 3, 10), (32, 7, 12), (32, 0, 4), (32, 31, 5)])
def test_ps_shout_right_shift_bit_exact(atlas, N, shift, log_T, mode):
    """RightShiftTable<N> read-raf (Sin / Cos trig downscale): device prover vs the oracle, both challenge modes."""
    import ctypes as C
    from oracle import orc, orc_ra as OR
    from jolt_atlas_amd import instances as I
    A = atlas
    orc.lib.fr_from_i64.argtypes = [C.c_int64, C.c_void_p]
    A.set_challenge_mode(mode); orc.lib.orc_set_challenge_mode(mode)
    try:
        T = 1 << log_T
        rng = np.random.default_rng(N * 100 + shift * 7 + log_T)
        idx = rng.integers(0, 1 << 20, size=T, dtype=np.uint64) if N == 32 else rng.integers(0, 1 << N, size=T, dtype=np.uint64)
        idx[0] = (1 << N) - 1
        if T > 2:
            idx[1] = 0; idx[2] = 1 << (N - 1)
        r_node, gamma = orc.random_fr(log_T, 5), orc.random_fr(1, 6)[0]
        # input claim: sum_t eq(r_node, t) ((idx_t >> shift) + gamma * signed(idx_t))
        E = orc.eq_evals(r_node)
        acc = orc.from_ints([0])[0]
        for t, k in enumerate(idx):
            k = int(k)
            x = k - (1 << N) if k >> (N - 1) else k
            w = orc.fr_array(1)
            orc.lib.fr_from_i64(x, orc._p(w))
            term = orc.fr_add_arr(orc.from_ints([k >> shift])[0], orc.fr_mul_arr(gamma, w[0]))
            acc = orc.fr_add_arr(acc, orc.fr_mul_arr(E[t], term))
        claim = acc
        t_o = orc.new_transcript(b"ps_rshift")
        o = OR.ps_rshift(idx, N, shift, r_node, gamma)
        rows_o, ch_o = o.prove(claim, t_o)
        inst = I.ps_shout_rshift(idx, N, shift, r_node, gamma)
        assert inst.num_rounds() == N + log_T and inst.degree() == 2
        t_g = A.Blake2bTranscript(b"ps_rshift")
        rows_g, ch_g = inst.prove(claim, t_g)
        assert ch_g == ch_o
        assert len(rows_g) == len(rows_o) and all(np.array_equal(a, b) for a, b in zip(rows_g, rows_o))
        assert t_g.state == t_o.state_bytes()
        if log_T <= 6:          # final claim: ra(r_address, r_cycle)
            rs = orc.challenges_to_fr(ch_g)
            r_addr, r_cyc = rs[:N], rs[N:][::-1]
            one = orc.from_ints([1])[0]
            vals = []
            for k in idx:
                w = one
                for i in range(N):
                    bit = (int(k) >> (N - 1 - i)) & 1
                    f = r_addr[i] if bit else orc.fr_add_arr(one, orc.fr_mul_arr(orc.from_ints([F_MINUS_ONE])[0], r_addr[i]))
                    w = orc.fr_mul_arr(w, f)
                vals.append(w)
            assert np.array_equal(inst.final_claims()[0], orc.evaluate(np.stack(vals), np.ascontiguousarray(r_cyc)))
        inst.free()
    finally:
        A.set_challenge_mode(0); orc.lib.orc_set_challenge_mode(0)


def test_ps_shout_rejects_unsupported_widths(atlas):
    from oracle import orc
    from jolt_atlas_amd import instances as I
    with pytest.raises(atlas.AtlasError):
        I.ps_shout_relu(np.zeros(4, dtype=np.uint64), 64, orc.random_fr(2, 1), orc.random_fr(1, 2)[0])
    with pytest.raises(atlas.AtlasError):
        I.ps_shout_relu(np.zeros(4, dtype=np.uint64), 8, orc.random_fr(2, 1), orc.random_fr(1, 2)[0])


@pytest.mark.parametrize("log_K,phases,log_T", [(8, 4, 2), (16, 8, 9), (16, 2, 5), (32, 8, 12), (64, 8, 7), (12, 3, 6), (20, 2, 6), (4, 4, 3)])
def test_identity_range_check_bit_exact(atlas, log_K, phases, log_T):
    from oracle import orc, orc_ra as OR
    from jolt_atlas_amd import instances as I
    A = atlas
    T = 1 << log_T
    rng = np.random.default_rng(log_K * 10 + log_T)
    hi = (1 << log_K) - 1
    idx = rng.integers(0, hi, size=T, dtype=np.uint64, endpoint=True)
    idx[0] = hi; idx[1] = 0
    r_node = orc.random_fr(log_T, 8)
    E = orc.eq_evals(r_node)
    claim = orc.fr_array(1)[0]
    for t in range(T):
        claim = orc.fr_add_arr(claim, orc.fr_mul_arr(E[t], orc.from_ints([int(idx[t])])[0]))
    t_o = orc.new_transcript(b"identity_rc")
    rows_o, ch_o = OR.ps_identity(idx, log_K, phases, r_node).prove(claim, t_o)
    inst = I.identity_range_check(idx, log_K, phases, r_node)
    assert inst.num_rounds() == log_K + log_T
    t_g = A.Blake2bTranscript(b"identity_rc")
    rows_g, ch_g = inst.prove(claim, t_g)
    assert ch_g == ch_o
    assert len(rows_g) == len(rows_o) and all(np.array_equal(a, b) for a, b in zip(rows_g, rows_o))
    assert t_g.state == t_o.state_bytes()
    assert len(inst.final_claims()) == 1
    inst.free()


@pytest.mark.parametrize("N,bound,sym,log_T", [(16, 9, True, 4), (32, 9, True, 10), (32, 17, True, 12), (32, 12, False, 9), (64, 31, True, 8), (16, 5, False, 1)])
def test_ps_shout_clamp_bit_exact(atlas, N, bound, sym, log_T):
    """ClampTable / ActivationClampTable / SoftmaxClampTable / SaturationTable shapes."""
    import ctypes as C
    from oracle import orc, orc_ra as OR
    from jolt_atlas_amd import instances as I
    A = atlas
    orc.lib.fr_from_i64.argtypes = [C.c_int64, C.c_void_p]
    T = 1 << log_T
    rng = np.random.default_rng(N + bound + log_T)
    vals = rng.integers(-(1 << (bound + 2)), 1 << (bound + 2), size=T, dtype=np.int64)
    vals[0] = -1; vals[1] = (1 << bound) - 1
    if T > 4:
        vals[2] = -(1 << bound); vals[3] = -(1 << (N - 1)) if N < 64 else -(1 << 62); vals[4] = (1 << (N - 2))
    idx = (vals.astype(np.int64).view(np.uint64) & np.uint64((1 << N) - 1 if N < 64 else 0xffffffffffffffff))
    r_node, gamma = orc.random_fr(log_T, 5), orc.random_fr(1, 6)[0]
    E = orc.eq_evals(r_node)
    claim = orc.fr_array(1)[0]
    lo = -(1 << bound) if sym else 0
    for t in range(T):
        x = int(vals[t])
        w = orc.fr_array(1); orc.lib.fr_from_i64(x, orc._p(w))
        rvv = orc.fr_array(1); orc.lib.fr_from_i64(min(max(x, lo), (1 << bound) - 1), orc._p(rvv))
        claim = orc.fr_add_arr(claim, orc.fr_mul_arr(E[t], orc.fr_add_arr(rvv[0], orc.fr_mul_arr(gamma, w[0]))))
    t_o = orc.new_transcript(b"ps_clamp")
    rows_o, ch_o = OR.ps_clamp(idx, N, bound, sym, r_node, gamma).prove(claim, t_o)
    inst = I.ps_shout_clamp(idx, N, bound, sym, r_node, gamma)
    assert inst.num_rounds() == N + log_T
    t_g = A.Blake2bTranscript(b"ps_clamp")
    rows_g, ch_g = inst.prove(claim, t_g)
    assert ch_g == ch_o
    assert len(rows_g) == len(rows_o) and all(np.array_equal(a, b) for a, b in zip(rows_g, rows_o))
    assert t_g.state == t_o.state_bytes()
    inst.free()


@pytest.mark.parametrize("log_T", [1, 5, 10, 13])
def test_ps_shout_unsigned_less_than_bit_exact(atlas, log_T):
    import ctypes as C
    from oracle import orc, orc_ra as OR
    from jolt_atlas_amd import instances as I
    A = atlas
    orc.lib.fr_from_i64.argtypes = [C.c_int64, C.c_void_p]
    T = 1 << log_T
    rng = np.random.default_rng(log_T)
    xs = rng.integers(0, 1 << 32, size=T, dtype=np.uint64)
    ys = rng.integers(0, 1 << 32, size=T, dtype=np.uint64)
    near = rng.random(T) < 0.5                       # divisions compare remainders with divisors: close operands
    ys[near] = xs[near] + rng.integers(-3, 4, size=int(near.sum())).astype(np.int64).astype(np.uint64)
    ys &= np.uint64(0xffffffff)
    xs[0], ys[0] = 5, 5
    xs[1], ys[1] = 0xffffffff, 0
    idx = np.array([OR.interleave(x, y) for x, y in zip(xs, ys)], dtype=np.uint64)
    r_node, gamma = orc.random_fr(log_T, 5), orc.random_fr(1, 6)[0]
    g2 = orc.fr_mul_arr(gamma, gamma)
    E = orc.eq_evals(r_node)
    sgn = lambda v: int(v) - (1 << 32) if int(v) >> 31 else int(v)
    claim = orc.fr_array(1)[0]
    for t in range(T):
        a, b = orc.fr_array(1), orc.fr_array(1)
        orc.lib.fr_from_i64(sgn(xs[t]), orc._p(a)); orc.lib.fr_from_i64(sgn(ys[t]), orc._p(b))
        term = orc.fr_add_arr(orc.from_ints([int(xs[t] < ys[t])])[0], orc.fr_add_arr(orc.fr_mul_arr(gamma, a[0]), orc.fr_mul_arr(g2, b[0])))
        claim = orc.fr_add_arr(claim, orc.fr_mul_arr(E[t], term))
    t_o = orc.new_transcript(b"ps_ult")
    rows_o, ch_o = OR.ps_ult(idx, r_node, gamma).prove(claim, t_o)
    inst = I.ps_shout_ult(idx, r_node, gamma)
    assert inst.num_rounds() == 64 + log_T and inst.degree() == 2
    t_g = A.Blake2bTranscript(b"ps_ult")
    rows_g, ch_g = inst.prove(claim, t_g)
    assert ch_g == ch_o
    assert len(rows_g) == len(rows_o) and all(np.array_equal(a, b) for a, b in zip(rows_g, rows_o))
    assert t_g.state == t_o.state_bytes()
    inst.free()
