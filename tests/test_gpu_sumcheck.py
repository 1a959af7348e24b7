"""GPU parity: the HIP sumcheck path (through the C-ABI) against the CPU oracle.

Bit-exact bar: every compressed round polynomial, every challenge, the final claims and
the transcript state must be identical (integer arithmetic; no tolerance)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _oracle_run(orc, L, R, eq, sched, a, b, label=b"synthetic_sc"):
    claim = orc.dot_claim(L, R, eq, sched, a, b)
    t = orc.new_transcript(label)
    proof, ch, fin = orc.sumcheck_dot_prove(L, R, claim, t, eq, sched, a, b)
    return claim, proof, ch, fin, t.state_bytes(), t.n_rounds


def _gpu_run(A, L, R, eq, sched, a, b, claim, label=b"synthetic_sc", i32=False, fs=None):
    n = len(L).bit_length() - 1
    if fs is not None:
        A.set_fs_mode(fs)
    mk = A.MultilinearPolynomial.from_i32 if i32 else A.MultilinearPolynomial.from_fr
    pl, pr = mk(L), mk(R)
    pe = A.MultilinearPolynomial.from_fr(eq) if eq is not None else None
    prover = A.EinsumDotProver(pl, pr, pe, sched, a, b)
    t = A.Blake2bTranscript(label)
    try:
        proof, ch, fin = A.Sumcheck.prove(prover, claim[0], t, n)
    finally:
        A.set_fs_mode(A.FS_HOST)
    prover.free()
    return proof, ch, fin, t.state, t.n_rounds


CASES = [
    # (n_vars, schedule, a, b)
    (1, 0, 0, 0), (2, 0, 0, 0), (5, 0, 0, 0), (10, 0, 0, 0), (11, 0, 0, 0), (12, 0, 0, 0), (14, 0, 0, 0),
    (16, 0, 0, 0),
    (6, 1, 3, 3), (6, 2, 3, 3), (5, 1, 0, 5), (7, 2, 7, 0), (7, 2, 0, 7), (7, 1, 7, 0),
    (13, 1, 5, 8), (13, 2, 5, 8), (13, 1, 12, 1), (13, 2, 1, 12), (14, 2, 10, 4), (14, 1, 2, 12),
]


@pytest.mark.parametrize("fs", [0, 1], ids=["fs_host", "fs_device"])
@pytest.mark.parametrize("n,sched,a,b", CASES)
def test_dot_sumcheck_bit_exact(atlas, n, sched, a, b, fs):
    from oracle import orc
    L = orc.random_fr(1 << n, 1000 + n)
    R = orc.random_fr(1 << n, 2000 + n)
    eq = None
    if sched:
        nb = a if sched == 1 else b
        eq = orc.eq_evals(orc.random_fr(nb, 3000 + n)) if nb else orc.from_ints([1])
    claim, proof_o, ch_o, fin_o, st_o, nr_o = _oracle_run(orc, L, R, eq, sched, a, b)
    proof_g, ch_g, fin_g, st_g, nr_g = _gpu_run(atlas, L, R, eq, sched, a, b, claim, fs=fs)
    assert ch_g == ch_o
    assert np.array_equal(proof_g, proof_o)
    assert np.array_equal(fin_g, fin_o)
    assert st_g == st_o and nr_g == nr_o
    assert orc.serialize_proof(proof_g) == orc.serialize_proof(proof_o)


@pytest.mark.parametrize("fs", [0, 1], ids=["fs_host", "fs_device"])
@pytest.mark.parametrize("n", [3, 10, 12, 15])
def test_dot_sumcheck_i32_operands(atlas, n, fs):
    """I32Scalars operands (|x| < 2^14, MODEL_SCALE activations): compact first round."""
    from oracle import orc
    rng = np.random.default_rng(n)
    L = rng.integers(-(1 << 14), 1 << 14, size=1 << n, dtype=np.int32)
    R = rng.integers(-(1 << 14), 1 << 14, size=1 << n, dtype=np.int32)
    L[0], R[0], L[1], R[1] = -(2 ** 31), 2 ** 31 - 1, 2 ** 31 - 1, -(2 ** 31)  # extremes
    Lf = orc.fr_array(1 << n); Rf = orc.fr_array(1 << n)
    orc.lib.orc_i32_to_fr(L.ctypes.data_as(orc.i32p), C.c_size_t(1 << n), orc._p(Lf))
    orc.lib.orc_i32_to_fr(R.ctypes.data_as(orc.i32p), C.c_size_t(1 << n), orc._p(Rf))
    claim = orc.dot_claim(Lf, Rf)
    t = orc.new_transcript(b"synthetic_sc")
    proof_o, ch_o, fin_o = orc.sumcheck_dot_prove_i32(L, R, claim, t)
    proof_g, ch_g, fin_g, st_g, nr_g = _gpu_run(atlas, L, R, None, 0, 0, 0, claim, i32=True, fs=fs)
    assert ch_g == ch_o
    assert np.array_equal(proof_g, proof_o)
    assert np.array_equal(fin_g, fin_o)
    assert st_g == t.state_bytes()


@pytest.mark.parametrize("n,sched,a,b", [(4, 0, 0, 0), (12, 0, 0, 0), (12, 1, 5, 7), (12, 2, 5, 7)])
def test_trait_shaped_round_api(atlas, n, sched, a, b):
    """compute_message / ingest_challenge driven from the host (the reference's own loop,
    sumcheck.rs:578-590) must give the same proof as the device-resident driver."""
    from oracle import orc
    A = atlas
    L = orc.random_fr(1 << n, 11 + n); R = orc.random_fr(1 << n, 12 + n)
    eq = None
    if sched:
        nb = a if sched == 1 else b
        eq = orc.eq_evals(orc.random_fr(nb, 13 + n))
    claim, proof_o, ch_o, fin_o, st_o, _ = _oracle_run(orc, L, R, eq, sched, a, b)
    prover = A.EinsumDotProver(A.MultilinearPolynomial.from_fr(L), A.MultilinearPolynomial.from_fr(R),
                               A.MultilinearPolynomial.from_fr(eq) if eq is not None else None, sched, a, b)
    t = orc.new_transcript(b"synthetic_sc")        # host transcript = the oracle's (checker side)
    orc.lib.orc_transcript_append_scalar(C.byref(t), orc._p(claim))
    prev = claim[0].copy()
    deg = 2 if sched == 0 else 3
    for rnd in range(n):
        coeffs = prover.compute_message(rnd, prev)
        assert coeffs.shape[0] == deg + 1
        cc = np.ascontiguousarray(np.concatenate([coeffs[:1], coeffs[2:]]))
        assert np.array_equal(cc, proof_o[rnd])
        orc.lib.orc_transcript_append_compressed(C.byref(t), orc._p(cc), C.c_size_t(deg))
        r = orc.fr_array(1)
        raw = (C.c_uint64 * 2)()
        orc.lib.orc_transcript_challenge_optimized(C.byref(t), raw, orc._p(r))
        c128 = raw[0] | (raw[1] << 64)
        assert c128 == ch_o[rnd]
        nxt = orc.fr_array(1)
        orc.lib.orc_unipoly_eval(orc._p(np.ascontiguousarray(coeffs)), C.c_size_t(deg + 1), orc._p(r), orc._p(nxt))
        prev = nxt[0]
        prover.ingest_challenge(c128, rnd)
    assert np.array_equal(prover.final_claims(), fin_o)
    prover.free()


def test_host_transcript_matches_oracle(atlas):
    from oracle import orc
    A = atlas
    t = A.Blake2bTranscript(b"ONNXProof")
    o = orc.new_transcript(b"ONNXProof")
    assert t.state == o.state_bytes()
    x = orc.random_fr(3, 5)
    t.append_message(b"hello"); orc.lib.orc_transcript_append_message(C.byref(o), b"hello")
    t.append_u64(0xdeadbeef12345678); orc.lib.orc_transcript_append_u64(C.byref(o), C.c_uint64(0xdeadbeef12345678))
    t.append_scalars(x); orc.lib.orc_transcript_append_scalars(C.byref(o), orc._p(x), C.c_size_t(3))
    t.append_bytes(b"\x01" * 200); orc.lib.orc_transcript_append_bytes(C.byref(o), (C.c_uint8 * 200)(*([1] * 200)), C.c_size_t(200))
    assert t.state == o.state_bytes()
    s = t.challenge_scalar(); so = orc.fr_array(1); orc.lib.orc_transcript_challenge_scalar(C.byref(o), orc._p(so))
    assert np.array_equal(s, so[0])
    assert t.state == o.state_bytes() and t.n_rounds == o.n_rounds


@pytest.mark.parametrize("n,sched,a,b", [(9, 0, 0, 0), (13, 0, 0, 0), (13, 1, 5, 8), (13, 2, 5, 8)])
def test_input_claim(atlas, n, sched, a, b):
    from oracle import orc
    A = atlas
    L = orc.random_fr(1 << n, 21 + n); R = orc.random_fr(1 << n, 22 + n)
    eq = None
    if sched:
        eq = orc.eq_evals(orc.random_fr(a if sched == 1 else b, 23 + n))
    prover = A.EinsumDotProver(A.MultilinearPolynomial.from_fr(L), A.MultilinearPolynomial.from_fr(R),
                               A.MultilinearPolynomial.from_fr(eq) if eq is not None else None, sched, a, b)
    assert np.array_equal(prover.input_claim(), orc.dot_claim(L, R, eq, sched, a, b)[0])
    prover.free()


def test_random_fr_generators_agree(atlas):
    from oracle import orc
    assert np.array_equal(atlas.random_fr(1000, 77), orc.random_fr(1000, 77))


def test_errors_are_loud(atlas):
    A = atlas
    with pytest.raises(A.AtlasError):
        A.MultilinearPolynomial.from_fr(np.zeros((3, 4), dtype=np.uint64))      # not a power of two
    p = A.MultilinearPolynomial.from_fr(np.zeros((4, 4), dtype=np.uint64))
    q = A.MultilinearPolynomial.from_fr(np.zeros((8, 4), dtype=np.uint64))
    with pytest.raises(A.AtlasError):
        A.EinsumDotProver(p, q)                                                 # length mismatch
    with pytest.raises(A.AtlasError):
        p.final_claim()                                                         # not fully bound
    p.free(); q.free()


@pytest.mark.parametrize("log_retained,log_reduced", [(3, 5), (1, 9), (6, 8)])
def test_mean_of_squares_mapping(atlas, log_retained, log_reduced):
    """MeanOfSquaresReductionProver = the dot prover with left = right = operand, EqSchedule::High (mean_of_squares.rs:363-398)."""
    from oracle import orc
    n = log_retained + log_reduced
    op = orc.random_fr(1 << n, 77 + n)
    eq = orc.eq_evals(orc.random_fr(log_retained, 78))
    claim, proof_o, ch_o, fin_o, st_o, nr_o = _oracle_run(orc, op, op.copy(), eq, 1, log_retained, log_reduced)
    proof_g, ch_g, fin_g, st_g, nr_g = _gpu_run(atlas, op, op.copy(), eq, 1, log_retained, log_reduced, claim)
    assert ch_g == ch_o and np.array_equal(proof_g, proof_o) and np.array_equal(fin_g, fin_o) and st_g == st_o
    assert np.array_equal(fin_g[0], fin_g[1])
