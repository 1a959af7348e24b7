"""GPU: one real operator node through the C-ABI — the fused-rescale Einsum (mk,kn->mn) as Einsum::prove composes it
(ops/einsum/mod.rs:71-115, fused_rebase.rs:215-285, clamp_lookups/mod.rs:264-309, shout.rs:399-466) — against the same
composition written over the ORACLE's instances: witness, every accumulator append, every challenge draw, the five
sumcheck proofs (bytes) and the final transcript state."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
FR = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001


def _one(orc):
    return orc.from_ints([1])[0]


def _eq_bits(orc, r, value, nbits):
    """prod_i (bit_i ? r_i : 1 - r_i), bit 0 of the product = MSB of value (big-endian point)."""
    w = _one(orc)
    minus1 = orc.from_ints([FR - 1])[0]
    for i in range(nbits):
        bit = (int(value) >> (nbits - 1 - i)) & 1
        f = r[i] if bit else orc.fr_add_arr(_one(orc), orc.fr_mul_arr(minus1, r[i]))
        w = orc.fr_mul_arr(w, f)
    return w


def _append(orc, t, x):
    orc.lib.orc_transcript_append_scalar(C.byref(t), orc._p(np.ascontiguousarray(x, dtype=np.uint64).reshape(1, 4)))


def _challenge_scalar(orc, t):
    s = orc.fr_array(1); orc.lib.orc_transcript_challenge_scalar(C.byref(t), orc._p(s))
    return s[0].copy()


def _challenge_opt(orc, t):
    raw = (C.c_uint64 * 2)(); r = orc.fr_array(1)
    orc.lib.orc_transcript_challenge_optimized(C.byref(t), raw, orc._p(r))
    return r[0].copy()


def _ser(orc, rows):
    from jolt_atlas_amd import wire
    return wire.sumcheck_proof_to_bytes(rows)


def _onehot_checks(orc, OR, OB, t, lookups, log_T, log_K, r_cycle, ra_point, ra_claim, claims):
    lkc = 4
    d = -(-log_K // lkc)
    T = 1 << log_T
    q = _challenge_scalar(orc, t)
    gp = [_one(orc)]
    for _ in range(1, d):
        gp.append(orc.fr_mul_arr(gp[-1], q))
    gp = np.stack(gp)
    gammas = np.stack([_challenge_opt(orc, t) for _ in range(d)])
    r_addr = np.stack([_challenge_opt(orc, t) for _ in range(lkc)])
    H = [((lookups >> np.uint64(lkc * (d - 1 - i))) & np.uint64(15)).astype(np.int32) for i in range(d)]
    G = OR.ra_G(H, lkc, r_cycle)
    pad = d * lkc - log_K
    chunks = np.concatenate([np.zeros((pad, 4), dtype=np.uint64), ra_point[:log_K]]).reshape(d, lkc, 4)
    r_cyc_ra = np.ascontiguousarray(ra_point[log_K:])
    hw_claim = orc.fr_array(1)[0]
    for x in gp:
        hw_claim = orc.fr_add_arr(hw_claim, x)
    zero = orc.fr_array(1)[0]
    insts = [OB.ra_instance(OR.ra_virtual(H, lkc, chunks, r_cyc_ra), ra_claim), OB.ra_instance(OR.hamming(G, lkc, gp), hw_claim),
             OB.ra_instance(OR.booleanity(G, H, lkc, gammas, r_addr, r_cycle), zero)]
    rows, ch, _ = OB.batched_prove(insts, t)
    rs = orc.challenges_to_fr(ch)
    mr = lkc + log_T
    # cache_openings, in instance order; each instance used the last num_rounds challenges
    ra_rs = np.ascontiguousarray(rs[mr - log_T:][::-1])
    for i in range(d):                                                   # RaVirtual: MLE of the gathered rows at the reversed challenges
        F = orc.eq_evals(chunks[i])
        c = orc.evaluate(np.ascontiguousarray(F[H[i]]), ra_rs)
        _append(orc, t, c); claims.append(c)
    hw_rs = np.ascontiguousarray(rs[mr - lkc:][::-1])
    for i in range(d):                                                   # HammingWeight: G_i at the reversed challenges
        c = orc.evaluate(G[i], hw_rs)
        _append(orc, t, c); claims.append(c)
    ba = np.ascontiguousarray(rs[:lkc][::-1]); bc = np.ascontiguousarray(rs[lkc:][::-1])
    Fb = orc.eq_evals(ba)
    for i in range(d):                                                   # Booleanity: H_i = eq(rho_address, idx_i(t)) at the reversed cycle challenges
        c = orc.evaluate(np.ascontiguousarray(Fb[H[i]]), bc)
        _append(orc, t, c); claims.append(c)
    return rows


def _fr_ints(orc, v):
    """Fr images of an integer vector: small values through the oracle's own i32 conversion, anything wider through Python integers"""
    v = np.asarray(v)
    if len(v) and np.abs(v.astype(np.int64)).max() < (1 << 31):
        from oracle import graph as OG
        return OG.fr_fast(v.astype(np.int32))
    return orc.from_ints([int(x) % FR for x in v])


# the last case is the smaller size bench.py times (T = 2^12, scale 2^14).  The bench's T = 2^16 shapes (16 x 1024 . 1024 x 4096: the k-sliced
# accumulation kernel with 64-bit atomics, k_ra_prod_f9 / _col at d = 16, the 96 KB-LDS Q build of the 64-bit clamp lookup; ReLU and Mul over
# 2^16 elements) run as one-operator graphs against COMMITTED oracle results in tests/test_gpu_graph_golden.py (node_einsum / node_relu /
# node_mul): the oracle side of those three cases cost the GPU box 117 s per run here.
# The oracle side of the two largest cases (15-16 s each on the GPU box) comes from committed digests (tests/golden/nodes_oracle.json, written by
# tests/golden/gen_nodes_oracle.py from the functions below); a tag that is not in the file is computed here.
def _node_digest(proofs, claims, state):
    import hashlib
    return {"proofs_sha256": [hashlib.sha256(bytes(p)).hexdigest() for p in proofs],
            "claims_sha256": hashlib.sha256(np.ascontiguousarray(claims).tobytes()).hexdigest(), "state": bytes(state).hex()}


def _node_expect(tag, run):
    import json, os
    p = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "nodes_oracle.json")
    if os.path.exists(p):
        doc = json.load(open(p))
        if tag in doc:
            return doc[tag]
    return _node_digest(*run())


def einsum_node_inputs(m, k, n, S):
    from oracle import orc
    rng = np.random.default_rng(m * 100 + k)
    lim = 1 << (6 if S < 14 else 14)                                   # bench size: activations and weights at scale 2^14
    A = rng.integers(-lim, lim, size=(m, k), dtype=np.int64).astype(np.int32)
    B = rng.integers(-lim, lim, size=(k, n), dtype=np.int64).astype(np.int32)
    T = m * n; log_T = T.bit_length() - 1
    return A, B, orc.random_fr(log_T, 77)


def einsum_node_oracle(m, k, n, S):
    """(serialized proofs, claims, final transcript state) of Einsum::prove with fused rescaling over the oracle's instances"""
    from oracle import orc, orc_ra as OR, orc_batched as OB
    A, B, r0 = einsum_node_inputs(m, k, n, S)
    T = m * n; log_T = T.bit_length() - 1; log_m = m.bit_length() - 1
    acc = (A.astype(np.int64) @ B.astype(np.int64)).reshape(-1)
    quot = acc >> S
    rem = acc - (quot << S)
    assert ((rem >= 0) & (rem < (1 << S))).all()
    outv = np.clip(quot, -(1 << 31), (1 << 31) - 1)
    f = lambda v: _fr_ints(orc, v)
    eval_R, acc_claim, out_claim = orc.evaluate(f(rem), r0), orc.evaluate(f(quot), r0), orc.evaluate(f(outv), r0)
    claims = []
    t = orc.new_transcript(b"einsum_node")
    _append(orc, t, eval_R); claims.append(eval_R)
    _append(orc, t, acc_claim); claims.append(acc_claim)
    gamma = _challenge_scalar(orc, t)
    cidx = quot.astype(np.int64).view(np.uint64).copy()
    exec_claim = orc.fr_add_arr(out_claim, orc.fr_mul_arr(gamma, acc_claim))
    rows_exec, ch = OR.ps_clamp(cidx, 64, 31, True, r0, gamma).prove(exec_claim, t)
    rs = orc.challenges_to_fr(ch)
    ra_point = np.concatenate([rs[:64], rs[64:][::-1]])
    ra_claim = OR.ra_claim(cidx, 64, ra_point)
    _append(orc, t, ra_claim); claims.append(ra_claim)
    rows_oh = _onehot_checks(orc, OR, OB, t, cidx, log_T, 64, r0, ra_point, ra_claim, claims)
    # matmul
    eq_m, eq_n = orc.eq_evals(np.ascontiguousarray(r0[:log_m])) if log_m else orc.from_ints([1]), orc.eq_evals(np.ascontiguousarray(r0[log_m:]))
    if m * k + k * n <= 4096:
        left = np.stack([sum_fr(orc, [orc.fr_mul_arr(f([A[i, l]])[0], eq_m[i]) for i in range(m)]) for l in range(k)])
        right = np.stack([sum_fr(orc, [orc.fr_mul_arr(f([B[l, j]])[0], eq_n[j]) for j in range(n)]) for l in range(k)])
    else:                                                            # MkKnMnLayout::fold through the oracle's C statement (pinned against the loop above at the small sizes)
        left, right = orc.fr_array(k), orc.fr_array(k)
        Ac, Bc = np.ascontiguousarray(A), np.ascontiguousarray(B)
        orc.lib.orc_fold_i32_cols(Ac.ctypes.data_as(C.c_void_p), C.c_size_t(m), C.c_size_t(k), orc._p(eq_m), orc._p(left))
        orc.lib.orc_fold_i32_rows(Bc.ctypes.data_as(C.c_void_p), C.c_size_t(k), C.c_size_t(n), orc._p(eq_n), orc._p(right))
    in_claim = orc.fr_add_arr(orc.fr_mul_arr(acc_claim, f([1 << S])[0]), eval_R)
    assert np.array_equal(orc.dot_claim(left, right)[0], in_claim)          # acc(r0) = rescaled(r0) 2^S + R(r0)
    proof_mm, ch_mm, fin_mm = orc.sumcheck_dot_prove(left, right, in_claim.reshape(1, 4), t)
    for c in (fin_mm[0], fin_mm[1]):
        _append(orc, t, c); claims.append(c)
    # remainder range check
    ridx = rem.astype(np.uint64)
    phases = 1 if S <= 2 else S // 4 if S % 4 == 0 else S // 2 if S % 2 == 0 else S
    rows_rc, ch = OR.ps_identity(ridx, S, phases, r0).prove(eval_R, t)
    rs = orc.challenges_to_fr(ch)
    rr_point = np.concatenate([rs[:S], rs[S:][::-1]])
    rr_claim = OR.ra_claim(ridx, S, rr_point)
    _append(orc, t, rr_claim); claims.append(rr_claim)
    rows_oh2 = _onehot_checks(orc, OR, OB, t, ridx, log_T, S, r0, rr_point, rr_claim, claims)
    want = [_ser(orc, rows_exec), _ser(orc, rows_oh), _ser(orc, [proof_mm[i] for i in range(len(proof_mm))]), _ser(orc, rows_rc), _ser(orc, rows_oh2)]
    return want, np.stack(claims), t.state_bytes()


@pytest.mark.parametrize("m,k,n,S", [(2, 8, 16, 6), (4, 4, 4, 4), (1, 16, 32, 7), (4, 64, 1024, 14)])
def test_einsum_node_matches_oracle_composition(atlas, m, k, n, S):
    from jolt_atlas_amd import node
    A_ = atlas
    A, B, r0 = einsum_node_inputs(m, k, n, S)
    want = _node_expect(f"einsum[{m}-{k}-{n}-{S}]", lambda: einsum_node_oracle(m, k, n, S))
    tA, tB = A_.TensorI32(A), A_.TensorI32(B)
    t_g = A_.Blake2bTranscript(b"einsum_node")
    proofs, claims_g, stage_ms = node.prove_einsum_node(tA, tB, m, k, n, S, r0, t_g)
    got = _node_digest(proofs, claims_g, t_g.state)
    for i, (a, b) in enumerate(zip(got["proofs_sha256"], want["proofs_sha256"])):
        assert a == b, f"proof {i} differs"
    assert got == want
    tA.free(); tB.free()


def sum_fr(orc, xs):
    s = orc.fr_array(1)[0]
    for x in xs:
        s = orc.fr_add_arr(s, x)
    return s


@pytest.mark.parametrize("log_T", [3, 6, 9, 12])
def test_relu_node_matches_oracle_composition(atlas, log_T):
    """ReLU::prove (ops/relu.rs:22-70) through atlas_prove_relu_node against the same composition over the oracle's instances:
    operand claim appended, gamma, PS-Shout over ReluTable<32>, its ra opening, the batched one-hot checks and their claims."""
    from oracle import orc, orc_ra as OR, orc_batched as OB
    from jolt_atlas_amd import node
    T = 1 << log_T
    rng = np.random.default_rng(log_T)
    x = rng.integers(-(1 << 12), 1 << 12, size=T, dtype=np.int64).astype(np.int32)
    x[0] = -(1 << 31); x[-1] = (1 << 31) - 1                                  # extremes of the i32 range
    out = np.maximum(x, 0)
    r0 = orc.random_fr(log_T, 99)
    f = lambda v: _fr_ints(orc, v)
    operand_claim, out_claim = orc.evaluate(f(x), r0), orc.evaluate(f(out), r0)
    claims = []
    t = orc.new_transcript(b"relu_node")
    _append(orc, t, operand_claim); claims.append(operand_claim)
    gamma = _challenge_scalar(orc, t)
    idx = x.astype(np.int64).astype(np.uint32).astype(np.uint64)             # `value as u32 as u64`
    exec_claim = orc.fr_add_arr(out_claim, orc.fr_mul_arr(gamma, operand_claim))
    rows_exec, ch = OR.ps_relu(idx, 32, r0, gamma).prove(exec_claim, t)
    rs = orc.challenges_to_fr(ch)
    ra_point = np.concatenate([rs[:32], rs[32:][::-1]])
    ra_claim = OR.ra_claim(idx, 32, ra_point)
    _append(orc, t, ra_claim); claims.append(ra_claim)
    rows_oh = _onehot_checks(orc, OR, OB, t, idx, log_T, 32, r0, ra_point, ra_claim, claims)
    tX = atlas.TensorI32(x)
    t_g = atlas.Blake2bTranscript(b"relu_node")
    proofs, claims_g, stage_ms = node.prove_relu_node(tX, log_T, r0, t_g)
    assert proofs[0] == _ser(orc, rows_exec), "execution proof differs"
    assert proofs[1] == _ser(orc, rows_oh), "one-hot proof differs"
    assert np.array_equal(claims_g, np.stack(claims))
    assert t_g.state == t.state_bytes()
    # the caller may hand the output claim over instead of having it evaluated
    t_g2 = atlas.Blake2bTranscript(b"relu_node")
    proofs2, claims_g2, _ = node.prove_relu_node(tX, log_T, r0, t_g2, output_claim=out_claim)
    assert proofs2 == proofs and t_g2.state == t_g.state
    tX.free()


def _fused_rescale_oracle(orc, OR, OB, label, acc, S, r0, inner):
    """The fused-rescale flow (prove_pre, the operator's sumcheck `inner(t, in_claim, claims) -> rows`, prove_remainder_rc)
    over the oracle's instances.  acc: the i64 accumulators.  Returns (list of 5 row lists, claims, transcript)."""
    T = len(acc); log_T = T.bit_length() - 1
    quot = acc >> S
    rem = acc - (quot << S)
    outv = np.clip(quot, -(1 << 31), (1 << 31) - 1)
    f = lambda v: _fr_ints(orc, v)
    eval_R, acc_claim, out_claim = orc.evaluate(f(rem), r0), orc.evaluate(f(quot), r0), orc.evaluate(f(outv), r0)
    claims = []
    t = orc.new_transcript(label)
    _append(orc, t, eval_R); claims.append(eval_R)
    _append(orc, t, acc_claim); claims.append(acc_claim)
    gamma = _challenge_scalar(orc, t)
    cidx = quot.astype(np.int64).view(np.uint64).copy()
    exec_claim = orc.fr_add_arr(out_claim, orc.fr_mul_arr(gamma, acc_claim))
    rows_exec, ch = OR.ps_clamp(cidx, 64, 31, True, r0, gamma).prove(exec_claim, t)
    rs = orc.challenges_to_fr(ch)
    ra_point = np.concatenate([rs[:64], rs[64:][::-1]])
    ra_claim = OR.ra_claim(cidx, 64, ra_point)
    _append(orc, t, ra_claim); claims.append(ra_claim)
    rows_oh = _onehot_checks(orc, OR, OB, t, cidx, log_T, 64, r0, ra_point, ra_claim, claims)
    in_claim = orc.fr_add_arr(orc.fr_mul_arr(acc_claim, f([1 << S])[0]), eval_R)
    rows_inner = inner(t, in_claim, claims)
    ridx = rem.astype(np.uint64)
    phases = 1 if S <= 2 else S // 4 if S % 4 == 0 else S // 2 if S % 2 == 0 else S
    rows_rc, ch = OR.ps_identity(ridx, S, phases, r0).prove(eval_R, t)
    rs = orc.challenges_to_fr(ch)
    rr_point = np.concatenate([rs[:S], rs[S:][::-1]])
    rr_claim = OR.ra_claim(ridx, S, rr_point)
    _append(orc, t, rr_claim); claims.append(rr_claim)
    rows_oh2 = _onehot_checks(orc, OR, OB, t, ridx, log_T, S, r0, rr_point, rr_claim, claims)
    return [rows_exec, rows_oh, rows_inner, rows_rc, rows_oh2], claims, t


def mul_node_inputs(log_T, S):
    from oracle import orc
    T = 1 << log_T
    rng = np.random.default_rng(100 + log_T)
    lim = 1 << (9 if S < 14 else 15)
    L = rng.integers(-lim, lim, size=T, dtype=np.int64).astype(np.int32)
    R = rng.integers(-lim, lim, size=T, dtype=np.int64).astype(np.int32)
    return L, R, orc.random_fr(log_T, 55)


def mul_node_oracle(log_T, S):
    """Mul::prove with fused rescaling (ops/mul.rs via impl_fused_rescale_proof_api) as the composition over the oracle's instances:
    MulProver between prove_pre and prove_remainder_rc"""
    from oracle import orc, orc_ra as OR, orc_batched as OB
    L, R, r0 = mul_node_inputs(log_T, S)
    acc = L.astype(np.int64) * R.astype(np.int64)
    f = lambda v: _fr_ints(orc, v)

    def inner(t, in_claim, claims):
        o = OR.elementwise(5, [f(L), f(R)], r0)                      # ATLAS_EW_MUL = MulProver
        assert np.array_equal(in_claim, orc.evaluate(f(acc), r0))    # acc(r0) = rescaled(r0) 2^S + R(r0)
        rows, _ch = o.prove(in_claim, t)
        fin = o.finals()
        for c in (fin[0], fin[1]):
            _append(orc, t, c); claims.append(c)
        return rows

    rows5, claims, t = _fused_rescale_oracle(orc, OR, OB, b"mul_node", acc, S, r0, inner)
    return [_ser(orc, rows) for rows in rows5], np.stack(claims), t.state_bytes()


@pytest.mark.parametrize("log_T,S", [(3, 5), (6, 7), (8, 4), (12, 14)])
def test_mul_node_matches_oracle_composition(atlas, log_T, S):
    """atlas_prove_mul_node against the oracle composition (mul_node_oracle)"""
    from jolt_atlas_amd import node
    L, R, r0 = mul_node_inputs(log_T, S)
    want = _node_expect(f"mul[{log_T}-{S}]", lambda: mul_node_oracle(log_T, S))
    tL, tR = atlas.TensorI32(L), atlas.TensorI32(R)
    t_g = atlas.Blake2bTranscript(b"mul_node")
    proofs, claims_g, stage_ms = node.prove_mul_node(tL, tR, log_T, S, r0, t_g)
    got = _node_digest(proofs, claims_g, t_g.state)
    for i, (a, b) in enumerate(zip(got["proofs_sha256"], want["proofs_sha256"])):
        assert a == b, f"proof {i} differs"
    assert got == want
    tL.free(); tR.free()


@pytest.mark.parametrize("log_T,S", [(4, 6), (7, 5)])
def test_square_node_matches_oracle_composition(atlas, log_T, S):
    """Square::prove with fused rescaling (ops/square.rs): SquareProver between prove_pre and prove_remainder_rc, one operand claim."""
    from oracle import orc, orc_ra as OR, orc_batched as OB
    from jolt_atlas_amd import node
    T = 1 << log_T
    x = np.random.default_rng(200 + log_T).integers(-(1 << 10), 1 << 10, size=T, dtype=np.int64).astype(np.int32)
    acc = x.astype(np.int64) * x.astype(np.int64)
    r0 = orc.random_fr(log_T, 56)
    f = lambda v: orc.from_ints([int(z) % FR for z in v])

    def inner(t, in_claim, claims):
        o = OR.elementwise(3, [f(x)], r0)                            # ATLAS_EW_SQUARE
        rows, _ch = o.prove(in_claim, t)
        c = o.finals()[0]
        _append(orc, t, c); claims.append(c)
        return rows

    rows5, claims, t = _fused_rescale_oracle(orc, OR, OB, b"square_node", acc, S, r0, inner)
    tX = atlas.TensorI32(x)
    t_g = atlas.Blake2bTranscript(b"square_node")
    proofs, claims_g, _ = node.prove_square_node(tX, log_T, S, r0, t_g)
    for i, (a, rows) in enumerate(zip(proofs, rows5)):
        assert a == _ser(orc, rows), f"proof {i} differs"
    assert np.array_equal(claims_g, np.stack(claims))
    assert t_g.state == t.state_bytes()
    tX.free()


@pytest.mark.parametrize("subtract", [False, True])
@pytest.mark.parametrize("log_T", [3, 7])
def test_addsub_node_matches_oracle_composition(atlas, log_T, subtract):
    """Add::prove / Sub::prove (ops/add.rs:70-105): prove_clamp_lookup over left +- right, then the operand tie; the extremes
    of the i32 range make the saturating clamp bite."""
    from oracle import orc, orc_ra as OR, orc_batched as OB
    from jolt_atlas_amd import node
    T = 1 << log_T
    rng = np.random.default_rng(300 + log_T)
    L = rng.integers(-(1 << 20), 1 << 20, size=T, dtype=np.int64).astype(np.int32)
    R = rng.integers(-(1 << 20), 1 << 20, size=T, dtype=np.int64).astype(np.int32)
    L[0], R[0] = (1 << 31) - 1, ((1 << 31) - 1) * (-1 if subtract else 1)          # overflows upwards
    L[1], R[1] = -(1 << 31), (1 << 31) - 1 if subtract else -(1 << 31)              # overflows downwards
    acc = L.astype(np.int64) - R.astype(np.int64) if subtract else L.astype(np.int64) + R.astype(np.int64)
    outv = np.clip(acc, -(1 << 31), (1 << 31) - 1)
    assert (outv != acc).any()
    r0 = orc.random_fr(log_T, 57)
    f = lambda v: orc.from_ints([int(z) % FR for z in v])
    acc_claim, out_claim = orc.evaluate(f(acc), r0), orc.evaluate(f(outv), r0)
    claims = []
    label = b"sub_node" if subtract else b"add_node"
    t = orc.new_transcript(label)
    _append(orc, t, acc_claim); claims.append(acc_claim)
    gamma = _challenge_scalar(orc, t)
    cidx = acc.astype(np.int64).view(np.uint64).copy()
    exec_claim = orc.fr_add_arr(out_claim, orc.fr_mul_arr(gamma, acc_claim))
    rows_exec, ch = OR.ps_clamp(cidx, 64, 31, True, r0, gamma).prove(exec_claim, t)
    rs = orc.challenges_to_fr(ch)
    ra_point = np.concatenate([rs[:64], rs[64:][::-1]])
    ra_claim = OR.ra_claim(cidx, 64, ra_point)
    _append(orc, t, ra_claim); claims.append(ra_claim)
    rows_oh = _onehot_checks(orc, OR, OB, t, cidx, log_T, 64, r0, ra_point, ra_claim, claims)
    for c in (orc.evaluate(f(L), r0), orc.evaluate(f(R), r0)):
        _append(orc, t, c); claims.append(c)
    tL, tR = atlas.TensorI32(L), atlas.TensorI32(R)
    t_g = atlas.Blake2bTranscript(label)
    proofs, claims_g, _ = node.prove_addsub_node(tL, tR, log_T, subtract, r0, t_g)
    assert proofs[0] == _ser(orc, rows_exec) and proofs[1] == _ser(orc, rows_oh)
    assert np.array_equal(claims_g, np.stack(claims))
    assert t_g.state == t.state_bytes()
    tL.free(); tR.free()


@pytest.mark.parametrize("op,n_ops", [(5, 2), (4, 3)])
@pytest.mark.parametrize("log_T", [3, 8])
def test_standard_node_matches_oracle_composition(atlas, op, n_ops, log_T):
    """And (MulProver without rescaling) / Iff through atlas_prove_standard_node: Sumcheck::prove with the node-output claim as
    the input claim, then the operand claims in input order (impl_standard_sumcheck_proof_api, ops/mod.rs:505-560)."""
    from oracle import orc, orc_ra as OR
    from jolt_atlas_amd import node
    T = 1 << log_T
    rng = np.random.default_rng(400 + log_T + op)
    f = lambda v: orc.from_ints([int(z) % FR for z in v])
    if op == 4:
        xs = [rng.integers(0, 2, size=T).astype(np.int32), rng.integers(-100, 100, size=T).astype(np.int32), rng.integers(-100, 100, size=T).astype(np.int32)]
        outv = np.where(xs[0] != 0, xs[1], xs[2])
    else:
        xs = [rng.integers(0, 2, size=T).astype(np.int32), rng.integers(0, 2, size=T).astype(np.int32)]
        outv = xs[0] * xs[1]
    r0 = orc.random_fr(log_T, 58)
    out_claim = orc.evaluate(f(outv), r0)
    t = orc.new_transcript(b"std_node")
    o = OR.elementwise(op, [f(x) for x in xs], r0)
    rows, _ch = o.prove(out_claim, t)
    claims = []
    for c in o.finals()[:n_ops]:
        _append(orc, t, c); claims.append(c)
    tens = [atlas.TensorI32(x) for x in xs]
    t_g = atlas.Blake2bTranscript(b"std_node")
    proof, claims_g = node.prove_standard_node(op, tens, log_T, r0, out_claim, t_g)
    assert proof == _ser(orc, rows)
    assert np.array_equal(claims_g, np.stack(claims))
    assert t_g.state == t.state_bytes()
    for x in tens:
        x.free()


def test_nodes_repeat_bit_for_bit(atlas):
    """Twelve runs of the ReLU and the Einsum node give one transcript state each: the lanes of the batched one-hot checks
    run on streams of their own, and an ordering slip between them and the library stream shows up as a changed proof
    (seen in bench.py before the lane streams were synchronised from the host)."""
    from jolt_atlas_amd import node as NODE
    A = atlas
    rng = np.random.default_rng(77)
    x = A.TensorI32(rng.integers(-(1 << 14), 1 << 14, size=1 << 12, dtype=np.int64).astype(np.int32))
    ta = A.TensorI32(rng.integers(-(1 << 14), 1 << 14, size=(4, 64), dtype=np.int64).astype(np.int32))
    tb = A.TensorI32(rng.integers(-(1 << 14), 1 << 14, size=(64, 256), dtype=np.int64).astype(np.int32))
    r12, r10 = A.random_fr(12, 5), A.random_fr(10, 6)
    relu, eins = set(), set()
    for _ in range(12):
        t = A.Blake2bTranscript(b"repeat_relu")
        NODE.prove_relu_node(x, 12, r12, t)
        relu.add(bytes(t.state))
        t = A.Blake2bTranscript(b"repeat_einsum")
        NODE.prove_einsum_node(ta, tb, 4, 64, 256, 14, r10, t)
        eins.add(bytes(t.state))
    x.free(); ta.free(); tb.free()
    assert len(relu) == 1 and len(eins) == 1


def test_node_does_not_read_uninitialised_device_memory():
    """The Einsum node in a child process whose device allocations are pre-filled (ATLAS_POOL_POISON, devpool.hpp) ends in the
    same transcript state as an unpoisoned child."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    states = set()
    for fill in (None, "165", "255"):
        env = dict(os.environ, PRINT_STATE="1", REPS="1")
        env.pop("ATLAS_POOL_POISON", None)
        if fill:
            env["ATLAS_POOL_POISON"] = fill
        out = subprocess.run([sys.executable, os.path.join(root, "tools", "time_node.py")], env=env, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        line = [ln for ln in out.stdout.splitlines() if ln.startswith("state ")]
        assert line, out.stdout[-2000:]
        states.add(line[-1])
    assert len(states) == 1, states
