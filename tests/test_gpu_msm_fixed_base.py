"""GPU parity for the fixed-base MSM path (atlas_srs_precompute + msm_tab_kernels.hip.h): with the table of
multiples 2^(c j) G_i in place every MSM, batch commit and HyperKZG::open must return the same points as the
oracle's plain Pippenger — byte equality of the affine image."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TAU_SEED = 0x51250001


def _tau(orc):
    return orc.random_fr(1, TAU_SEED)[0]


@pytest.fixture(scope="module")
def srs_ref(atlas):
    from oracle import orc
    n = 1 << 11
    return orc.srs_powers(_tau(orc), n)


# (window_bits, q): digit width c' = window_bits / q.  8/1: one bin per vector; 16/1: 8 bins; 16/2: two bucket sets;
# 20/1: 128 bins; 21/3: three sets of 7-bit digits; 24/1: 2048 bins, 2^23 buckets
SHAPES = [(8, 1), (16, 1), (16, 2), (20, 1), (20, 2), (21, 3), (24, 1)]


@pytest.mark.parametrize("wb,q", SHAPES)
@pytest.mark.parametrize("n", [1, 3, 100, 1000, 2048])
def test_fixed_base_msm_matches_oracle(atlas, srs_ref, monkeypatch, wb, q, n):
    from oracle import orc
    monkeypatch.setenv("ATLAS_MSM_TAB_Q", str(q))
    s = atlas.SRS.upload(srs_ref)
    info = s.precompute(0, wb)
    assert info == {"n_points": len(srs_ref), "window_bits": wb, "levels": (255 + wb - 1) // wb}
    sc = orc.random_fr(n, 900 + n)
    assert orc.g1_eq(s.msm(sc), orc.msm(srs_ref[:n], sc))
    if n >= 100:
        assert orc.g1_eq(s.msm(sc[:n - 37], offset=37), orc.msm(srs_ref[37:n], sc[:n - 37]))
    s.free()


@pytest.mark.parametrize("wb,q", [(16, 1), (20, 2)])
@pytest.mark.parametrize("first,count", [(512, 512), (1024, 1024), (0, 512), (37, 1000)])
def test_table_over_a_point_range(atlas, srs_ref, monkeypatch, wb, q, first, count):
    """atlas_srs_precompute_range: the table over the powers [first, first + count) only — a rank's share in a sharded proof.  MSMs whose
    points lie inside the range go through it, the others take the variable-base path: every one equals the oracle's point."""
    from oracle import orc
    monkeypatch.setenv("ATLAS_MSM_TAB_Q", str(q))
    s = atlas.SRS.upload(srs_ref)
    info = s.precompute_range(first, count, wb)
    assert info["n_points"] == count and info["window_bits"] == wb
    sc = orc.random_fr(count, 4242 + first)
    assert orc.g1_eq(s.msm(sc, offset=first), orc.msm(srs_ref[first:first + count], sc))                       # exactly the range
    assert orc.g1_eq(s.msm(sc[:count // 2], offset=first + 5), orc.msm(srs_ref[first + 5:first + 5 + count // 2], sc[:count // 2]))   # inside it
    n_out = min(300, len(srs_ref) - first - count + 100)
    if first + count - 100 + n_out <= len(srs_ref):                                                              # straddling its end: no table
        assert orc.g1_eq(s.msm(sc[:n_out], offset=first + count - 100), orc.msm(srs_ref[first + count - 100:first + count - 100 + n_out], sc[:n_out]))
    assert orc.g1_eq(s.msm(sc[:200]), orc.msm(srs_ref[:200], sc[:200]))                                          # from 0: inside only when first == 0
    s.free()


@pytest.mark.parametrize("wb,q", [(16, 1), (20, 1), (16, 2)])
def test_fixed_base_edge_scalars(atlas, srs_ref, monkeypatch, wb, q):
    """zero / one / r-1 / powers of two around the digit boundaries; all-equal scalars (every entry of a digit in one
    bucket: the workgroup-per-bucket reduction); all-zero input; points at infinity in the SRS."""
    from oracle import orc
    from oracle.pymodel import field as F
    monkeypatch.setenv("ATLAS_MSM_TAB_Q", str(q))
    c = wb // q
    s = atlas.SRS.upload(srs_ref)
    s.precompute(0, wb)
    vals = [0, 1, F.FR - 1, 2, (1 << 128) - 1, 1 << 253, 0, 0, 12345, F.FR - 2 ** 15, 1 << (c - 1), (1 << (c - 1)) - 1,
            (1 << (c - 1)) + 1, (1 << c) - 1, 1 << c, (1 << (2 * c)) - 1, (1 << 254) % F.FR, 7, 0, F.FR - (1 << (c - 1))]
    sc = orc.from_ints(vals)
    assert orc.g1_eq(s.msm(sc), orc.msm(srs_ref[:len(vals)], sc, naive=True))
    assert int(s.msm(orc.from_ints([0] * 64))["infinity"]) == 1
    same = np.repeat(orc.random_fr(1, 3), 1500, axis=0)
    assert orc.g1_eq(s.msm(same), orc.msm(srs_ref[:1500], same))
    small = orc.from_ints([int(v) for v in np.random.default_rng(1).integers(0, 4, size=2048)])
    assert orc.g1_eq(s.msm(small), orc.msm(srs_ref, small))
    s.free()
    with_inf = srs_ref[:256].copy()
    for i in (0, 5, 255):
        with_inf[i]["x"] = 0; with_inf[i]["y"] = 0; with_inf[i]["infinity"] = 1
    s = atlas.SRS.upload(with_inf)
    s.precompute(0, wb)
    sc = orc.random_fr(256, 41)
    assert orc.g1_eq(s.msm(sc), orc.msm(with_inf, sc))
    s.free()


def test_table_prefix_and_replacement(atlas, srs_ref, monkeypatch):
    """A table over a prefix serves the MSMs inside it; longer ones take the variable-base path; a second
    precompute replaces the first; ATLAS_MSM_TAB=0 ignores the table."""
    from oracle import orc
    monkeypatch.setenv("ATLAS_MSM_TAB_Q", "1")
    s = atlas.SRS.upload(srs_ref)
    assert s.table_info()["window_bits"] == 0
    s.precompute(1024, 16)
    sc = orc.random_fr(2048, 77)
    assert orc.g1_eq(s.msm(sc[:1024]), orc.msm(srs_ref[:1024], sc[:1024]))
    assert orc.g1_eq(s.msm(sc), orc.msm(srs_ref, sc))
    assert orc.g1_eq(s.msm(sc[:600], offset=500), orc.msm(srs_ref[500:1100], sc[:600]))      # crosses the table's end
    assert s.precompute(0, 12)["window_bits"] == 12
    assert orc.g1_eq(s.msm(sc), orc.msm(srs_ref, sc))
    monkeypatch.setenv("ATLAS_MSM_TAB", "0")
    assert orc.g1_eq(s.msm(sc), orc.msm(srs_ref, sc))
    with pytest.raises(atlas.AtlasError):
        s.precompute(0, 30)
    s.free()


@pytest.mark.parametrize("wb,q", [(16, 1), (16, 2), (20, 1)])
def test_fixed_base_batch_commit_and_open(atlas, monkeypatch, wb, q):
    """CommitmentScheme::batch_commit and HyperKZG::open with the table: same commitments, witnesses, evaluations
    and transcript as the oracle."""
    from oracle import orc
    A = atlas
    monkeypatch.setenv("ATLAS_MSM_TAB_Q", str(q))
    ell = 11
    n = 1 << ell
    tau = _tau(orc)
    srs_g = A.SRS.generate(tau, n)
    srs_g.precompute(0, wb)
    srs_o = srs_g.download()
    polys = [orc.random_fr(m, 300 + m) for m in (n, n // 2, 256, n)]
    dev = [A.MultilinearPolynomial.from_fr(p) for p in polys]
    got = srs_g.commit_batch(dev)
    for g_, p in zip(got, polys):
        assert orc.g1_eq(g_, orc.msm(srs_o[:len(p)], p))
    rng = np.random.default_rng(ell)
    pt = [int(rng.integers(0, 1 << 62)) << 66 | int(rng.integers(0, 1 << 62)) for _ in range(ell)]
    t_o = orc.new_transcript(b"TestEval")
    com_o, w_o, v_o = orc.hyperkzg_open(srs_o, polys[0], pt, t_o)
    t_g = A.Blake2bTranscript(b"TestEval")
    com_g, w_g, v_g = A.HyperKZG.open(srs_g, dev[0], pt, t_g)
    assert np.array_equal(v_g, v_o)
    assert all(orc.g1_eq(x, y) for x, y in zip(com_g, com_o)) and all(orc.g1_eq(x, y) for x, y in zip(w_g, w_o))
    assert t_g.state == t_o.state_bytes()
    for d in dev:
        d.free()
    srs_g.free()


@pytest.mark.parametrize("log_n", [16, 19])
def test_fixed_base_large_trapdoor_identity(atlas, log_n):
    """Size-independent check with the library's own choice of table and digit width: with bases tau^(i+1) G the MSM
    equals (sum s_i tau^(i+1)) G; and the table path agrees with the variable-base path on the same input."""
    from oracle import orc
    n = 1 << log_n
    tau = _tau(orc)
    s = atlas.SRS.generate(tau, n)
    sc = orc.random_fr(n, 177 + log_n)
    plain = s.msm(sc)
    s.precompute()
    got = s.msm(sc)
    k = orc.fr_array(1)
    t = np.ascontiguousarray(tau).reshape(1, 4)
    orc.lib.orc_eval_as_univariate(orc._p(np.ascontiguousarray(sc)), C.c_size_t(n), orc._p(t), orc._p(k))
    k = orc.fr_mul_arr(k[0], tau)
    assert orc.g1_eq(got, orc.g1_mul_generator(k))
    assert orc.g1_eq(got, plain)
    s.free()


@pytest.mark.parametrize("table", [False, True])
@pytest.mark.parametrize("hi", [2, 64, 1 << 13])
def test_msm_of_small_field_elements(atlas, table, hi):
    """LargeScalars polynomials that hold small integers: every entry of digit 0 lands in a few buckets (one bucket
    with 2^16 entries for hi = 2), which exercises the 16-lane and the two-stage bucket reductions.  Checked through
    the trapdoor identity MSM = (sum s_i tau^(i+1)) G."""
    from oracle import orc
    n = 1 << 16
    tau = _tau(orc)
    s = atlas.SRS.generate(tau, n)
    if table:
        s.precompute(0, 16)
    vals = np.random.default_rng(hi).integers(1 if hi == 2 else 0, hi, size=n)
    sc = orc.from_ints([int(v) for v in vals])
    got = s.msm(sc)
    k = orc.fr_array(1)
    t = np.ascontiguousarray(tau).reshape(1, 4)
    orc.lib.orc_eval_as_univariate(orc._p(np.ascontiguousarray(sc)), C.c_size_t(n), orc._p(t), orc._p(k))
    k = orc.fr_mul_arr(k[0], tau)
    assert orc.g1_eq(got, orc.g1_mul_generator(k))
    s.free()
