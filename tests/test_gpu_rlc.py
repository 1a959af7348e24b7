"""GPU parity: build_materialized_rlc (poly/rlc_polynomial.rs:13-78) vs the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _onehot(T, K, seed, none_frac=0.1):
    rng = np.random.default_rng(seed)
    k = rng.integers(0, K, size=T, dtype=np.int32)
    k[rng.random(T) < none_frac] = -1
    return k


@pytest.mark.parametrize("case", [
    dict(dense=[("fr", 12)], onehot=[]),
    dict(dense=[("fr", 10), ("i32", 12), ("fr", 12), ("i32", 7)], onehot=[]),
    dict(dense=[], onehot=[(1 << 8, 16), (1 << 8, 16), (1 << 6, 16)]),
    dict(dense=[("fr", 14), ("i32", 13)], onehot=[(1 << 10, 16), (1 << 10, 16), (1 << 9, 16), (1 << 12, 4), (1 << 4, 256)]),
    dict(dense=[("i32", 3)], onehot=[(1 << 12, 16)]),
])
def test_rlc_matches_oracle(atlas, case):
    from oracle import orc, orc_batched as OB
    from jolt_atlas_amd import rlc
    A = atlas
    dense_o, dense_g, polys = [], [], []
    for j, (kind, n) in enumerate(case["dense"]):
        c = orc.random_fr(1, 50 + j)[0]
        if kind == "fr":
            a = orc.random_fr(1 << n, 60 + j)
            p = A.MultilinearPolynomial.from_fr(a)
        else:
            a = np.random.default_rng(70 + j).integers(-(1 << 31), (1 << 31) - 1, size=1 << n, dtype=np.int32)
            a[:4] = [0, -(1 << 31), (1 << 31) - 1, -1]
            p = A.MultilinearPolynomial.from_i32(a)
        polys.append(p); dense_o.append((a, c)); dense_g.append((p, c))
    oh = [(_onehot(T, K, 80 + j), K, orc.random_fr(1, 90 + j)[0]) for j, (T, K) in enumerate(case["onehot"])]
    want = OB.rlc_build(dense_o, oh)
    joint = rlc.build_materialized_rlc(dense_g, oh)
    got = joint.to_host()
    assert got.shape == want.shape
    assert np.array_equal(got, want)
    # inputs are not consumed
    for p, (a, _) in zip(polys, dense_o):
        if a.dtype != np.int32:
            assert np.array_equal(p.to_host(), a)
        p.free()
    joint.free()


def test_rlc_rejects_non_power_of_two(atlas):
    from oracle import orc
    from jolt_atlas_amd import rlc
    with pytest.raises(atlas.AtlasError):
        rlc.build_materialized_rlc([], [(_onehot(100, 16, 1), 16, orc.random_fr(1, 2)[0])])
