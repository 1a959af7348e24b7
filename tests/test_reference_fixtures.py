"""Replays fixtures exported from a run of the reference (tools/export_ref_fixtures.rs) through the oracle (CPU) and the
device (GPU).  The reference cannot be built in this image (Rust), so the fixture is absent here and the tests are EXPECTED FAILURES (xfail) with
a loud reason; with tests/golden/ref_fixtures.json in place they pin the oracle — and the three encoding inferences of
SURVEY.md App. A — against the reference's own bytes in one step."""
import ctypes as C
import json
import os

import numpy as np
import pytest

PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_fixtures.json")
FR = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001
FQ = 0x30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47

SELF = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "selfcheck_fixtures.json")
UNPINNED = ("PARITY UNPINNED: tests/golden/ref_fixtures.json is absent — export it from the reference with "
            "tools/export_ref_fixtures.rs (needs cargo; cannot run in this image) to pin the oracle against the reference")

# "reference" = bytes exported from the reference (the pin; an expected failure — `x`, "xfailed" in the summary — while absent: the missing pin shows in every run);
# "selfcheck" = the same layout generated from this repository's oracle (gen_selfcheck_fixture.py): pins nothing, keeps the
# replay code below exercised in every session
needs_fixture = pytest.mark.parametrize("fx", ["reference", "selfcheck"], indirect=True)


@pytest.fixture(scope="module")
def fx(request):
    if request.param == "reference":
        if not os.path.exists(PATH):
            pytest.xfail(UNPINNED)
        return json.load(open(PATH))
    return json.load(open(SELF))


def _fr_list(orc, hx):
    b = bytes.fromhex(hx)
    return orc.from_ints([int.from_bytes(b[i:i + 32], "little") for i in range(0, len(b), 32)])


def _fr_hex(orc, a):
    """canonical LE hex of one Montgomery Fr (the oracle's fr_to_bytes_le)."""
    out = (C.c_uint8 * 32)()
    orc.lib.fr_to_bytes_le(orc._p(np.ascontiguousarray(a, dtype=np.uint64)), out)
    return bytes(out).hex()


def _u128(hx):
    return int.from_bytes(bytes.fromhex(hx), "little")


def _decompress(hx):
    """ark compressed G1Affine -> (x, y) integers or None for infinity."""
    b = bytearray(bytes.fromhex(hx)); flags = b[31] & 0xc0; b[31] &= 0x3f
    if flags & 0x40:
        return None
    x = int.from_bytes(b, "little")
    y = pow(x ** 3 + 3, (FQ + 1) // 4, FQ)
    assert y * y % FQ == (x ** 3 + 3) % FQ
    if (y > FQ - y) != bool(flags & 0x80):
        y = FQ - y
    return x, y


@needs_fixture
def test_ref_transcript_history(fx):
    """every transcript operation reproduces the reference's state (blake2b.rs:81-238)."""
    from oracle import orc
    t = None
    for op in fx["transcript"]:
        k = op["op"]
        if k == "new":
            t = orc.new_transcript(bytes.fromhex(op["arg"]))
        elif k == "append_message":
            m = bytes.fromhex(op["arg"]); orc.lib.orc_transcript_append_message(C.byref(t), m)
        elif k == "append_u64":
            orc.lib.orc_transcript_append_u64(C.byref(t), C.c_uint64(int.from_bytes(bytes.fromhex(op["arg"]), "little")))
        elif k == "append_scalar":
            orc.lib.orc_transcript_append_scalar(C.byref(t), orc._p(_fr_list(orc, op["arg"])))
        elif k == "append_scalars":
            v = _fr_list(orc, op["arg"]); orc.lib.orc_transcript_append_scalars(C.byref(t), orc._p(v), C.c_size_t(len(v)))
        elif k == "append_point":
            xy = _decompress(op["arg"])
            buf = bytes(64) if xy is None else xy[0].to_bytes(32, "big") + xy[1].to_bytes(32, "big")      # blake2b.rs:166-187
            orc.lib.orc_transcript_append_bytes(C.byref(t), (C.c_uint8 * 64)(*buf), C.c_size_t(64))
        elif k == "challenge_u128":
            raw = (C.c_uint64 * 2)(); orc.lib.orc_transcript_challenge_u128(C.byref(t), raw)
            assert raw[0] | (raw[1] << 64) == _u128(op["out"])
        elif k == "challenge_scalar":
            s = orc.fr_array(1); orc.lib.orc_transcript_challenge_scalar(C.byref(t), orc._p(s))
            assert _fr_hex(orc, s[0]) == op["out"]
        elif k == "challenge_scalar_optimized":
            raw = (C.c_uint64 * 2)(); r = orc.fr_array(1)
            orc.lib.orc_transcript_challenge_optimized(C.byref(t), raw, orc._p(r))
            assert (raw[0] | (raw[1] << 64)) & ((1 << 125) - 1) == _u128(op["out_u128"])
            assert _fr_hex(orc, r[0]) == op["out"], "MontU128Challenge -> Fr reading (SURVEY App. A.2) differs from the reference"
        else:
            raise AssertionError("unknown op " + k)
        assert t.state_bytes().hex() == op["state"], f"transcript state after {k}"


@needs_fixture
def test_ref_challenge_values(fx):
    """MontU128Challenge::from(x) as a field element, and challenge * Fr (mont_ark_u128.rs:51-92)."""
    from oracle import orc
    for row in fx["challenge_to_fr"]:
        x = _u128(row["u128"])
        f = orc.challenges_to_fr([x])[0]
        assert _fr_hex(orc, f) == row["fr"]
    cm = fx["challenge_mul"]
    a = _fr_list(orc, cm["a"])[0]
    c = orc.challenges_to_fr([_u128(cm["u128"])])[0]
    assert _fr_hex(orc, orc.fr_mul_arr(a, c)) == cm["product"]


def _check_sumcheck(fx, orc, proof, ch, fin, state):
    sc = fx["sumcheck"]
    n = sc["n"]
    for i in range(n):
        want = bytes.fromhex(sc["compressed_polys"][i])
        assert int.from_bytes(want[:8], "little") == 2
        got = b"".join(bytes.fromhex(_fr_hex(orc, proof[i][k])) for k in range(2))
        assert got == want[8:], f"round {i} polynomial"
        assert ch[i] & ((1 << 125) - 1) == _u128(sc["challenges"][i])
    assert _fr_hex(orc, fin[0]) == sc["final_left"] and _fr_hex(orc, fin[1]) == sc["final_right"]
    assert state.hex() == sc["state"]


@needs_fixture
def test_ref_sumcheck_oracle(fx):
    from oracle import orc
    sc = fx["sumcheck"]
    L, R = _fr_list(orc, sc["left"]), _fr_list(orc, sc["right"])
    claim = orc.dot_claim(L, R)
    assert _fr_hex(orc, claim[0]) == sc["claim"]
    t = orc.new_transcript(b"synthetic_sc")
    proof, ch, fin = orc.sumcheck_dot_prove(L, R, claim, t)
    _check_sumcheck(fx, orc, proof, ch, fin, t.state_bytes())


@needs_fixture
@pytest.mark.gpu
def test_ref_sumcheck_device(fx, atlas):
    from oracle import orc
    A = atlas
    sc = fx["sumcheck"]
    L, R = _fr_list(orc, sc["left"]), _fr_list(orc, sc["right"])
    for fs in (A.FS_HOST, A.FS_DEVICE):
        A.set_fs_mode(fs)
        try:
            prover = A.EinsumDotProver(A.MultilinearPolynomial.from_fr(L), A.MultilinearPolynomial.from_fr(R))
            claim = prover.input_claim()
            t = A.Blake2bTranscript(b"synthetic_sc")
            proof, ch, fin = A.Sumcheck.prove(prover, claim, t, sc["n"])
            prover.free()
        finally:
            A.set_fs_mode(A.FS_HOST)
        _check_sumcheck(fx, orc, proof, ch, fin, t.state)


@needs_fixture
@pytest.mark.gpu
def test_ref_hyperkzg_device(fx, atlas):
    """commitment, proof bytes (ark serialize_compressed) and transcript state of HyperKZG::open at ell = 4, through the C-ABI:
    SRS uploaded from the fixture's compressed g1_powers, proof serialized by atlas_hyperkzg_proof_serialize."""
    from oracle import orc
    from jolt_atlas_amd import wire
    A = atlas
    hk = fx["hyperkzg"]
    ell = hk["ell"]
    raw = bytes.fromhex(hk["g1_powers"])
    n_pts = len(raw) // 32
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "srs.bin")
        open(p, "wb").write(n_pts.to_bytes(8, "little") + raw + bytes(16))
        srs = wire.srs_load_file(p)
    poly = A.MultilinearPolynomial.from_fr(_fr_list(orc, hk["poly"]))
    point = [_u128(x) for x in hk["point"]]
    com = A.HyperKZG.commit(srs, poly)
    assert wire.g1_to_bytes(com).hex() == hk["commitment"]
    t = A.Blake2bTranscript(b"TestEval")
    c, w, v = A.HyperKZG.open(srs, poly, point, t)
    assert wire.hyperkzg_proof_to_bytes(c, w, v).hex() == hk["proof"]
    assert t.state.hex() == hk["state"]
    poly.free(); srs.free()


# ---- whole ONNXProof::prove runs (tools/export_ref_graph_fixtures.rs): operator compositions + the proof container --------------------
GPATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_graph_fixtures.json")
GSELF = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "selfcheck_graph_fixtures.json")
GUNPINNED = ("PARITY UNPINNED: tests/golden/ref_graph_fixtures.json is absent — export it from the reference with "
             "tools/export_ref_graph_fixtures.rs (needs cargo; cannot run in this image) to pin the operator compositions and the proof container")
needs_graph_fixture = pytest.mark.parametrize("gfx", ["reference", "selfcheck"], indirect=True)


@pytest.fixture(scope="module")
def gfx(request):
    if request.param == "reference":
        if not os.path.exists(GPATH):
            pytest.xfail(GUNPINNED)
        return json.load(open(GPATH))
    return json.load(open(GSELF))


def _g1_array(orc, hexes):
    """compressed ark points -> the oracle's / the library's affine Montgomery layout"""
    out = np.zeros(len(hexes), dtype=orc.G1_DTYPE)
    for i, hx in enumerate(hexes):
        xy = _decompress(hx)
        if xy is None:
            out[i]["infinity"] = 1
            continue
        for name, v in zip(("x", "y"), xy):
            m = v * (1 << 256) % FQ
            out[i][name] = [(m >> (64 * k)) & ((1 << 64) - 1) for k in range(4)]
    return out


def _graph_of(g):
    nodes = []
    for nd in g["nodes"]:
        d = dict(nd)
        if "data" in d:
            d["data"] = np.asarray(d["data"], dtype=np.int32)
        nodes.append(d)
    return nodes, g["outputs"], [np.asarray(a, dtype=np.int32) for a in g["inputs"]]


@needs_graph_fixture
def test_ref_graph_proofs_oracle(gfx):
    """oracle/graph.py over the file's SRS powers reproduces the file's ONNXProof bytes (serialize_proof) and output tensor"""
    from oracle import graph as OG, orc
    for g in gfx["graphs"]:
        nodes, outputs, inputs = _graph_of(g)
        P = OG.Prover(nodes, outputs, _g1_array(orc, g["srs_g1"]))
        proof = P.prove(inputs)
        assert [int(x) for x in P.trace[outputs[0]]] == g["output"], g["name"]
        assert proof.hex() == g["proof"], "ONNXProof bytes of " + g["name"]


@needs_graph_fixture
@pytest.mark.gpu
def test_ref_graph_proofs_device(gfx, atlas):
    """atlas_prove_graph over the file's SRS powers reproduces the file's ONNXProof bytes; atlas_verify_graph accepts them"""
    from oracle import orc
    from jolt_atlas_amd import graph as GG
    for g in gfx["graphs"]:
        nodes, outputs, inputs = _graph_of(g)
        srs = atlas.SRS.upload(_g1_array(orc, g["srs_g1"]))
        G = GG.Graph(nodes, outputs)
        proof, _state, _tm = G.prove(srs, inputs)
        assert [int(x) for x in G.node_output(outputs[0])] == g["output"], g["name"]
        assert proof.hex() == g["proof"], "ONNXProof bytes of " + g["name"]
        G.free(); srs.free()


def test_fixture_status_is_reported():
    """always runs: states in the test log whether the oracle is pinned against the reference."""
    if os.path.exists(GPATH):
        print("reference graph fixtures present: operator compositions and the proof container pinned against the reference")
    else:
        print("PARITY UNPINNED: no reference graph fixtures (tools/export_ref_graph_fixtures.rs has not been run)")
    if os.path.exists(PATH):
        print("reference fixtures present: oracle pinned against the reference")
    else:
        print("PARITY UNPINNED: no reference fixtures (tools/export_ref_fixtures.rs has not been run)")
