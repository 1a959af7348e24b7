"""softmax_last_axis two-phase provers and SumAxis on the device vs the oracle (bit-exact rows, challenges, transcript,
final claims)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

KINDS = {"exp_sum": 0, "max": 1, "recip": 2, "sum_axis": 3}


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("kind,log_K,log_N,as_i32", [
    ("exp_sum", 1, 1, True), ("exp_sum", 3, 4, False), ("exp_sum", 6, 8, True), ("exp_sum", 4, 12, False), ("exp_sum", 10, 6, True),
    ("max", 1, 1, True), ("max", 3, 4, False), ("max", 6, 8, True), ("max", 4, 12, True),
    ("recip", 1, 1, True), ("recip", 3, 4, False), ("recip", 6, 8, True), ("recip", 4, 12, False), ("recip", 9, 7, True),
    ("sum_axis", 0, 1, True), ("sum_axis", 0, 9, False), ("sum_axis", 0, 16, True),
])
def test_softmax_instances_bit_exact(atlas, kind, log_K, log_N, as_i32, mode):
    from oracle import orc, orc_ra as OR
    from jolt_atlas_amd import instances as I
    A = atlas
    A.set_challenge_mode(mode); orc.lib.orc_set_challenge_mode(mode)
    try:
        code = KINDS[kind]
        n, K = 1 << (log_K + log_N), 1 << log_K
        rng = np.random.default_rng(1000 * code + 10 * log_K + log_N)

        def make(count, lo, hi, seed):
            if as_i32:
                v = rng.integers(lo, hi, size=count, dtype=np.int64)
                return v, orc.from_ints([int(x) for x in v])
            return None, orc.random_fr(count, seed)
        ai, af = make(n, -(1 << 20), 1 << 20, 11)
        bi = bf = None
        if kind == "max":
            bi = np.zeros(n, dtype=np.int64)
            bi[(np.arange(K) << log_N) + rng.integers(0, 1 << log_N, size=K)] = 1
            bf = orc.from_ints([int(x) for x in bi])
            if not as_i32:
                bi = None
        if kind == "recip":
            bi, bf = make(K, 1, 1 << 20, 12)
        r = orc.random_fr(log_K + (log_N if kind == "recip" else 0), 13) if kind != "sum_axis" else None
        claim = orc.random_fr(1, 14)[0]
        o = OR.softmax(code, af, bf, log_K, log_N, r)
        t_o = orc.new_transcript(b"sm")
        rows_o, ch_o = o.prove(claim, t_o)
        mk = lambda vi, vf: A.MultilinearPolynomial.from_i32(vi.astype(np.int32)) if vi is not None else A.MultilinearPolynomial.from_fr(vf)
        pa = mk(ai, af)
        pb = mk(bi, bf) if bf is not None else None
        inst = I.softmax_instance(code, pa, pb, log_K, log_N, r)
        assert inst.num_rounds() == log_K + log_N
        t_g = A.Blake2bTranscript(b"sm")
        rows_g, ch_g = inst.prove(claim, t_g)
        assert ch_g == ch_o
        assert len(rows_g) == len(rows_o) and all(np.array_equal(x, y) for x, y in zip(rows_g, rows_o))
        assert t_g.state == t_o.state_bytes()
        assert np.array_equal(np.stack(inst.final_claims()), o.finals())
        inst.free(); pa.free()
        if pb is not None:
            pb.free()
    finally:
        A.set_challenge_mode(0); orc.lib.orc_set_challenge_mode(0)


def test_softmax_instance_errors(atlas):
    from oracle import orc
    from jolt_atlas_amd import instances as I
    A = atlas
    p = A.MultilinearPolynomial.from_fr(orc.random_fr(16, 1))
    q = A.MultilinearPolynomial.from_fr(orc.random_fr(4, 2))
    with pytest.raises(A.AtlasError):
        I.softmax_instance(0, p, q, 2, 2, orc.random_fr(2, 3))        # ExpSum takes one operand
    with pytest.raises(A.AtlasError):
        I.softmax_instance(1, p, q, 2, 2, orc.random_fr(2, 3))        # MaxIndicator: e must have the full length
    with pytest.raises(A.AtlasError):
        I.softmax_instance(2, p, p, 2, 2, orc.random_fr(4, 3))        # RecipMult: inv_sum has 2^log_K entries
    with pytest.raises(A.AtlasError):
        I.softmax_instance(3, p, None, 1, 3, None)                    # SumAxis: log_K = 0
    with pytest.raises(A.AtlasError):
        I.softmax_instance(0, p, None, 2, 3, orc.random_fr(2, 3))     # length mismatch
    p.free(); q.free()
