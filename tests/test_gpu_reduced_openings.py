"""GPU parity: ONNXProof::prove_reduced_openings (prover.rs:141-176) — batched opening-reduction
sumcheck, claim/gamma exchange, joint RLC polynomial, HyperKZG opening — vs the same pipeline composed
from the oracle's parts, plus acceptance by the oracle's trapdoor HyperKZG verifier."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _onehot(T, K, seed):
    rng = np.random.default_rng(seed)
    k = rng.integers(0, K, size=T).astype(np.int32)
    k[rng.random(T) < 0.1] = -1
    return k


@pytest.mark.parametrize("shape", [
    dict(dense=[10], onehot=[]),
    dict(dense=[10, 10, 7], onehot=[(4, 6)]),
    dict(dense=[12], onehot=[(4, 8), (4, 8), (4, 5)]),
    dict(dense=[], onehot=[(4, 7), (2, 9)]),
    dict(dense=[11], onehot=[(4, 7)] * 5, shared_cycle=True),      # a lookup op's d polynomials: one r_cycle, own r_address each
    dict(dense=[], onehot=[(4, 6), (4, 6), (2, 8), (4, 6)], shared_cycle=True),
])
def test_prove_reduced_openings_bit_exact(atlas, shape):
    from oracle import orc, orc_ra as OR, orc_batched as OB
    from jolt_atlas_amd import reduced
    A = atlas
    max_vars = max(shape["dense"] + [a + b for a, b in shape["onehot"]])
    tau = orc.random_fr(1, 0x51250001)[0]
    srs_h = orc.srs_powers(tau, 1 << max_vars)
    srs = A.SRS.generate(tau, 1 << max_vars)
    openings_g, inst_o, claims, rlc_dense, rlc_oh, kinds = [], [], [], [], [], []
    polys = []
    for j, n in enumerate(shape["dense"]):
        p, pt = orc.random_fr(1 << n, 10 + j), orc.random_fr(n, 20 + j)
        c = orc.evaluate(p, pt)
        dp = A.MultilinearPolynomial.from_fr(p); polys.append(dp)
        openings_g.append(dict(poly=dp, point=pt, claim=c))
        inst_o.append(OB.ra_instance(OR.dense_opening(p, pt), c)); claims.append(c); kinds.append(("d", p))
    for j, (log_K, log_T) in enumerate(shape["onehot"]):
        k = _onehot(1 << log_T, 1 << log_K, 30 + j)
        ra, rc = orc.random_fr(log_K, 40 + j), orc.random_fr(log_T, 50 + (0 if shape.get("shared_cycle") else j))
        Fa = orc.eq_evals(ra)
        vec = np.stack([Fa[x] if x >= 0 else np.zeros(4, dtype=np.uint64) for x in k])
        c = orc.evaluate(vec, rc)
        openings_g.append(dict(k=k, log_K=log_K, r_address=ra, r_cycle=rc, claim=c))
        inst_o.append(OB.ra_instance(OR.onehot_opening(k, log_K, ra, rc), c)); claims.append(c); kinds.append(("o", k, 1 << log_K))
    # ---- oracle pipeline (prover.rs:141-176 composed from its parts)
    t_o = orc.new_transcript(b"reduced_openings")
    rows_o, ch_o, _ = OB.batched_prove(inst_o, t_o)
    fin_o = []                       # sumcheck_claims: P_i at its slice of r_sumcheck (opening_reduction.rs:238-246)
    rs = orc.challenges_to_fr(ch_o)
    for kd in kinds:
        if kd[0] == "d":
            n = int(np.log2(len(kd[1])))
            fin_o.append(orc.evaluate(kd[1], np.ascontiguousarray(rs[len(rs) - n:])))
        else:
            k, K = kd[1], kd[2]
            log_K, log_T = int(np.log2(K)), int(np.log2(len(k)))
            sl = rs[len(rs) - log_K - log_T:]
            Fs = orc.eq_evals(np.ascontiguousarray(sl[:log_K]))
            vec = np.stack([Fs[x] if x >= 0 else np.zeros(4, dtype=np.uint64) for x in k])
            fin_o.append(orc.evaluate(vec, np.ascontiguousarray(sl[log_K:])))
    fin_o = np.stack(fin_o)
    orc.lib.orc_transcript_append_scalars(C.byref(t_o), orc._p(np.ascontiguousarray(fin_o)), C.c_size_t(len(fin_o)))
    q = orc.fr_array(1); orc.lib.orc_transcript_challenge_scalar(C.byref(t_o), orc._p(q))
    gam = [orc.from_ints([1])[0]]
    for _ in range(1, len(fin_o)):
        gam.append(orc.fr_mul_arr(gam[-1], q[0]))
    dense_o = [(kd[1], g) for kd, g in zip(kinds, gam) if kd[0] == "d"]
    oh_o = [(kd[1], kd[2], g) for kd, g in zip(kinds, gam) if kd[0] == "o"]
    joint = OB.rlc_build(dense_o, oh_o)
    assert len(joint) == 1 << max_vars
    t_before_open = type(t_o).from_buffer_copy(t_o)
    com_o, w_o, v_o = orc.hyperkzg_open(srs_h, joint, ch_o, t_o)
    # ---- device
    t_g = A.Blake2bTranscript(b"reduced_openings")
    rows_g, ch_g, claims_g, com_g, w_g, v_g = reduced.prove_reduced_openings(openings_g, srs, t_g)
    assert ch_g == ch_o
    assert len(rows_g) == len(rows_o) and all(np.array_equal(a, b) for a, b in zip(rows_g, rows_o))
    assert np.array_equal(claims_g, fin_o)
    assert all(orc.g1_eq(a, b) for a, b in zip(com_g, com_o)) and all(orc.g1_eq(a, b) for a, b in zip(w_g, w_o))
    assert np.array_equal(v_g, v_o)
    assert t_g.state == t_o.state_bytes()
    # the dense polynomials are intact
    for dp, kd in zip(polys, [k for k in kinds if k[0] == "d"]):
        assert np.array_equal(dp.to_host(), kd[1])
        dp.free()
    # the joint opening verifies against C = commit(joint), y = joint(r_sumcheck) (hyperkzg/mod.rs:451-509)
    Cj = orc.msm(srs_h, joint)
    y = orc.evaluate(joint, np.ascontiguousarray(rs))
    tv = A.Blake2bTranscript(b"x")                       # the product verifier from the same transcript state, through the pairing
    for i in range(32):
        tv.t.state[i] = t_before_open.state[i]
    tv.t.n_rounds = t_before_open.n_rounds
    assert orc.hyperkzg_verify_trapdoor(srs_h, tau, Cj, ch_g, y, com_g, w_g, v_g, t_before_open)
    vk = A.HyperKZG.vk_from_trapdoor(tau, srs.download(0, 1)[0])
    assert A.HyperKZG.verify(vk, Cj, ch_g, y, com_g, w_g, v_g, tv) and tv.state == t_g.state
    srs.free()


def test_accumulator_matches_manual_composition(atlas):
    """ProverOpeningAccumulator::{append_dense, append_sparse, append_virtual} (opening_proof.rs:265-409) + the reduced-openings
    stage through atlas_accumulator_* = the same transcript appends done by hand + atlas_prove_reduced_openings; the openings come
    back out of the map; keys order the committed polynomials like BTreeMap<CommittedPoly>."""
    from oracle import orc
    from jolt_atlas_amd import reduced
    A = atlas
    tau = orc.random_fr(1, 0x51250001)[0]
    srs = A.SRS.generate(tau, 1 << 11)
    p0, pt0 = orc.random_fr(1 << 10, 1), orc.random_fr(10, 2)
    p1, pt1 = orc.random_fr(1 << 8, 3), orc.random_fr(8, 4)
    c0, c1 = orc.evaluate(p0, pt0), orc.evaluate(p1, pt1)
    log_K, log_T = 4, 7
    ks = [_onehot(1 << log_T, 1 << log_K, 60 + j) for j in range(3)]
    ra, rc = orc.random_fr(log_K, 7), orc.random_fr(log_T, 8)
    Fa = orc.eq_evals(ra)
    cs = [orc.evaluate(np.stack([Fa[x] if x >= 0 else np.zeros(4, dtype=np.uint64) for x in k]), rc) for k in ks]
    vpt, vclaim = orc.random_fr(5, 9), orc.random_fr(1, 10)[0]
    d0, d1 = A.MultilinearPolynomial.from_fr(p0), A.MultilinearPolynomial.from_fr(p1)
    # accumulator: poly keys chosen so that CommittedPoly order = one-hot rows 0..2 (keys 1..3), then p1 (5), then p0 (9);
    # appended in another order
    acc = reduced.ProverOpeningAccumulator()
    t_a = A.Blake2bTranscript(b"acc")
    acc.append_dense(t_a, 100, 9, d0, pt0, c0)
    acc.append_virtual(t_a, 50, vpt, vclaim)
    acc.append_sparse(t_a, [201, 202, 203], [1, 2, 3], ks, log_K, ra, rc, cs)
    acc.append_dense(t_a, 101, 5, d1, pt1, c1)
    got_pt, got_c = acc.get(50)
    assert np.array_equal(got_pt, vpt) and np.array_equal(got_c, vclaim)
    got_pt, got_c = acc.get(202)
    assert np.array_equal(got_pt, np.concatenate([ra, rc])) and np.array_equal(got_c, cs[1])
    with pytest.raises(A.AtlasError):
        acc.get(999)
    out_a = acc.prove_reduced_openings(srs, t_a)
    # by hand
    t_m = A.Blake2bTranscript(b"acc")
    for c in (c0, vclaim, *cs, c1):
        t_m.append_scalar(c)
    ops = [dict(k=k, log_K=log_K, r_address=ra, r_cycle=rc, claim=c) for k, c in zip(ks, cs)]
    ops += [dict(poly=d1, point=pt1, claim=c1), dict(poly=d0, point=pt0, claim=c0)]
    out_m = reduced.prove_reduced_openings(ops, srs, t_m)
    assert out_a[1] == out_m[1] and all(np.array_equal(a, b) for a, b in zip(out_a[0], out_m[0]))
    assert np.array_equal(out_a[2], out_m[2]) and np.array_equal(out_a[5], out_m[5])
    assert all(orc.g1_eq(a, b) for a, b in zip(out_a[3], out_m[3])) and all(orc.g1_eq(a, b) for a, b in zip(out_a[4], out_m[4]))
    assert t_a.state == t_m.state
    acc.free(); d0.free(); d1.free(); srs.free()
