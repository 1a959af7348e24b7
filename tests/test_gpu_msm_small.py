"""GPU parity: VariableBaseMSM::msm on the narrow MultilinearPolynomial variants
(joltworks/src/msm/mod.rs:38-181) through atlas_msm_small / atlas_msm_poly, vs the oracle's
restatement (pos-MSM minus neg-MSM with arkworks-style Pippenger)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TAU_SEED = 0x51250001


@pytest.fixture(scope="module")
def srs(atlas):
    from oracle import orc
    n = 1 << 13
    tau = orc.random_fr(1, TAU_SEED)[0]
    s = atlas.SRS.generate(tau, n)
    yield s, orc.srs_powers(tau, n)
    s.free()


def _draw(dtype, n, seed, lo=None, hi=None):
    rng = np.random.default_rng(seed)
    info = np.iinfo(dtype)
    lo = info.min if lo is None else lo
    hi = info.max if hi is None else hi
    return rng.integers(lo, hi, size=n, dtype=dtype, endpoint=True)


@pytest.mark.parametrize("dtype", [np.uint8, np.uint16, np.uint32, np.uint64, np.int32, np.int64])
@pytest.mark.parametrize("n", [1, 7, 256, 1000, 8192])
def test_msm_small_full_range(atlas, srs, dtype, n):
    from oracle import orc, orc_batched as OB
    s, ref = srs
    sc = _draw(dtype, n, 11 * n + np.dtype(dtype).itemsize)
    assert orc.g1_eq(s.msm_small(sc), OB.msm_small(ref[:n], sc))


def test_msm_small_edges(atlas, srs):
    from oracle import orc, orc_batched as OB
    s, ref = srs
    for dtype in (np.int32, np.int64):
        info = np.iinfo(dtype)
        sc = np.array([info.min, info.max, 0, -1, 1, info.min + 1, 0, 2, -2, info.min], dtype=dtype)
        assert orc.g1_eq(s.msm_small(sc), OB.msm_small(ref[:len(sc)], sc))
    for dtype in (np.uint8, np.uint16, np.uint32, np.uint64):
        info = np.iinfo(dtype)
        sc = np.array([info.max, 0, 1, info.max - 1, info.max, 2 ** (info.bits - 1), 2 ** (info.bits - 1) - 1], dtype=dtype)
        assert orc.g1_eq(s.msm_small(sc), OB.msm_small(ref[:len(sc)], sc))
    # all zero -> infinity (msm/mod.rs:41-47); binary scalars (msm_binary branch, :48-59)
    z = s.msm_small(np.zeros(100, dtype=np.uint8))
    assert int(z["infinity"]) == 1
    b = _draw(np.uint8, 3000, 5, 0, 1)
    assert orc.g1_eq(s.msm_small(b), OB.msm_small(ref[:3000], b))
    # cancels to infinity
    sc = np.array([5, -5], dtype=np.int32)
    two = np.concatenate([ref[:1], ref[:1]])
    s2 = atlas.SRS.upload(two)
    assert int(s2.msm_small(sc)["infinity"]) == 1
    s2.free()


@pytest.mark.parametrize("bits", [1, 4, 14, 20])
def test_msm_small_narrow_values_balanced_plan(atlas, srs, bits):
    """Values far narrower than their type (activations are |x| < 2^14 in i32): the plan follows
    the data, results unchanged; matches the Fr path on the same values."""
    from oracle import orc, orc_batched as OB
    s, ref = srs
    n = 8192
    sc = _draw(np.int32, n, bits, -(1 << bits) + 1, (1 << bits) - 1)
    got = s.msm_small(sc)
    assert orc.g1_eq(got, OB.msm_small(ref[:n], sc))
    fr = orc.fr_array(n)
    import ctypes as C
    orc.lib.orc_i32_to_fr(sc.ctypes.data_as(orc.i32p), C.c_size_t(n), orc._p(fr))
    assert orc.g1_eq(got, s.msm(fr))
    # I32Scalars polynomial resident on the device: atlas_msm_poly takes the same plan
    p = atlas.MultilinearPolynomial.from_i32(sc)
    assert orc.g1_eq(s.msm(p), got)
    p.free()


def test_msm_small_offset_and_length_error(atlas, srs):
    from oracle import orc, orc_batched as OB
    s, ref = srs
    sc = _draw(np.uint16, 500, 3)
    assert orc.g1_eq(s.msm_small(sc, offset=100), OB.msm_small(ref[100:600], sc))
    with pytest.raises(atlas.AtlasError):
        s.msm_small(_draw(np.uint8, 8193, 1))
