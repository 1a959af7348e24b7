"""Wire formats: ark-serialize images through the C-ABI against an independent Python statement of the encodings
(SURVEY App. A.3), round trips, malformed input, the 368-byte ell = 2 HyperKZG proof size the reference's own test pins
(hyperkzg/tests.rs:107-109), and the SRS file (hyperkzg/mod.rs:60-95) decompressed on the device."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FQ = 0x30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47
FR = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001


def _int(limbs):
    return sum(int(x) << (64 * i) for i, x in enumerate(limbs))


def _canon_fq(mont):
    return _int(mont) * pow(1 << 256, -1, FQ) % FQ


def _canon_fr(mont):
    return _int(mont) * pow(1 << 256, -1, FR) % FR


def _py_compress(pt):
    if pt["infinity"]:
        return bytes(31) + b"\x40"
    x, y = _canon_fq(pt["x"]), _canon_fq(pt["y"])
    b = bytearray(x.to_bytes(32, "little"))
    if y > FQ - y:
        b[31] |= 0x80
    return bytes(b)


def test_field_and_point_encodings(atlas):
    from oracle import orc
    from jolt_atlas_amd import wire
    for f in list(orc.random_fr(20, 1)) + [orc.from_ints([0])[0], orc.from_ints([1])[0], orc.from_ints([FR - 1])[0]]:
        b = wire.fr_to_bytes(f)
        assert b == _canon_fr(f).to_bytes(32, "little")
        assert np.array_equal(wire.fr_from_bytes(b), f)
    with pytest.raises(atlas.AtlasError):
        wire.fr_from_bytes(FR.to_bytes(32, "little"))                 # not reduced
    srs = atlas.SRS.generate(orc.random_fr(1, 0x51250001)[0], 64)
    pts = srs.download()
    for p_ in pts:
        b = wire.g1_to_bytes(p_)
        assert b == _py_compress(p_)
        q = wire.g1_from_bytes(b)
        assert np.array_equal(q["x"], p_["x"]) and np.array_equal(q["y"], p_["y"]) and not q["infinity"]
        u = wire.g1_to_bytes(p_, compressed=False)
        assert u[:32] == _canon_fq(p_["x"]).to_bytes(32, "little") and (u[63] & 0x3f) == (_canon_fq(p_["y"]) >> 248) & 0x3f
        assert (u[63] >> 7) == (b[31] >> 7)
    inf = np.zeros(1, dtype=atlas.G1_DTYPE)[0]; inf["infinity"] = 1
    assert wire.g1_to_bytes(inf) == bytes(31) + b"\x40"
    assert wire.g1_from_bytes(bytes(31) + b"\x40")["infinity"]
    for bad in (bytes(31) + b"\xc0", b"\x01" + bytes(30) + b"\x40", FQ.to_bytes(32, "little"), (4).to_bytes(32, "little")):
        # both flags; x with the infinity flag; x >= p; x = 4 is not on y^2 = x^3 + 3 (67 is a non-residue mod p)
        with pytest.raises(atlas.AtlasError):
            wire.g1_from_bytes(bad)
    srs.free()


def test_sumcheck_and_hyperkzg_proof_bytes(atlas):
    from oracle import orc
    from jolt_atlas_amd import wire
    A = atlas
    n = 6
    L, R = orc.random_fr(1 << n, 1), orc.random_fr(1 << n, 2)
    claim = orc.dot_claim(L, R)
    prover = A.EinsumDotProver(A.MultilinearPolynomial.from_fr(L), A.MultilinearPolynomial.from_fr(R))
    proof, ch, fin = A.Sumcheck.prove(prover, claim[0], A.Blake2bTranscript(b"wire"), n)
    prover.free()
    rows = [proof[i] for i in range(n)]
    b = wire.sumcheck_proof_to_bytes(rows)
    assert b == orc.serialize_proof(proof)                             # the oracle's statement of the same struct
    assert len(b) == 8 + n * (8 + 2 * 32)
    back, used = wire.sumcheck_proof_from_bytes(b)
    assert used == len(b) and all(np.array_equal(x, y) for x, y in zip(back, rows))
    with pytest.raises(A.AtlasError):
        wire.sumcheck_proof_from_bytes(b[:-1])
    # HyperKZG proof at ell = 2: 368 bytes (hyperkzg/tests.rs:107-109)
    for ell in (2, 5):
        tau = orc.random_fr(1, 0x51250001)[0]
        srs = A.SRS.generate(tau, 1 << ell)
        poly = A.MultilinearPolynomial.from_fr(orc.random_fr(1 << ell, 7))
        point = [int.from_bytes(np.random.default_rng(ell + i).bytes(16), "little") for i in range(ell)]
        com, w, v = A.HyperKZG.open(srs, poly, point, A.Blake2bTranscript(b"TestEval"))
        pb = wire.hyperkzg_proof_to_bytes(com, w, v)
        assert len(pb) == 8 + 32 * (ell - 1) + 8 + 96 + 8 + 3 * (8 + 32 * ell)
        if ell == 2:
            assert len(pb) == 368
        c2, w2, v2 = wire.hyperkzg_proof_from_bytes(pb)
        assert all(orc.g1_eq(a, b_) for a, b_ in zip(c2, com)) and all(orc.g1_eq(a, b_) for a, b_ in zip(w2, w))
        assert np.array_equal(v2, np.asarray(v).reshape(3, ell, 4))
        assert wire.hyperkzg_proof_to_bytes(c2, w2, v2) == pb
        poly.free(); srs.free()


def test_srs_file_round_trip(atlas, tmp_path):
    from oracle import orc
    from jolt_atlas_amd import wire
    A = atlas
    srs = A.SRS.generate(orc.random_fr(1, 0x51250001)[0], 3000)
    path = tmp_path / "srs.bin"
    wire.srs_save_file(srs, path)
    raw = path.read_bytes()
    pts = srs.download()
    assert len(raw) == 8 + 32 * 3000 + 16 and int.from_bytes(raw[:8], "little") == 3000
    assert raw[8:40] == _py_compress(pts[0]) and raw[8 + 32 * 2999:8 + 32 * 3000] == _py_compress(pts[2999])
    back = wire.srs_load_file(path)
    got = back.download()
    assert np.array_equal(got["x"], pts["x"]) and np.array_equal(got["y"], pts["y"])
    part = wire.srs_load_file(path, 100)
    assert len(part.download()) == 100
    # a corrupted point is refused
    bad = bytearray(raw); bad[8 + 32 * 17] ^= 1
    (tmp_path / "bad.bin").write_bytes(bytes(bad))
    # flipping a bit of x lands off the curve with probability 1/2; try a few until one does
    refused = False
    for k in range(8):
        bad = bytearray(raw); bad[8 + 32 * 17 + 1] ^= (1 << k)
        (tmp_path / "bad.bin").write_bytes(bytes(bad))
        try:
            wire.srs_load_file(tmp_path / "bad.bin").free()
        except A.AtlasError:
            refused = True
            break
    assert refused
    srs.free(); back.free(); part.free()
