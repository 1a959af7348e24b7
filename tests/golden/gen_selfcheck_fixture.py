"""Writes tests/golden/selfcheck_fixtures.json: the SAME layout tools/export_ref_fixtures.rs produces from the reference, but
generated from this repository's oracle.  It pins nothing (oracle against oracle); it exists so that the replay code of
tests/test_reference_fixtures.py runs in every test session and is known to work the day a real reference fixture arrives.
Run from the repository root:  python tests/golden/gen_selfcheck_fixture.py"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import orc  # noqa: E402

FQ = 0x30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47


def fr_hex(a):
    out = (C.c_uint8 * 32)()
    orc.lib.fr_to_bytes_le(orc._p(np.ascontiguousarray(a, dtype=np.uint64)), out)
    return bytes(out).hex()


def fq_canon(m):
    return sum(int(x) << (64 * i) for i, x in enumerate(m)) * pow(1 << 256, -1, FQ) % FQ


def g1_hex(p):
    if int(p["infinity"]):
        return (bytes(31) + b"\x40").hex()
    x, y = fq_canon(p["x"]), fq_canon(p["y"])
    b = bytearray(x.to_bytes(32, "little"))
    if y > FQ - y:
        b[31] |= 0x80
    return bytes(b).hex()


def u128_hex(x):
    return int(x).to_bytes(16, "little").hex()


out = {}
# transcript
ops = []
t = orc.new_transcript(b"ref_fixture")
ops.append(dict(op="new", arg=b"ref_fixture".hex(), state=t.state_bytes().hex()))
orc.lib.orc_transcript_append_message(C.byref(t), b"hello")
ops.append(dict(op="append_message", arg=b"hello".hex(), state=t.state_bytes().hex()))
orc.lib.orc_transcript_append_u64(C.byref(t), C.c_uint64(0xdeadbeef12345678))
ops.append(dict(op="append_u64", arg=(0xdeadbeef12345678).to_bytes(8, "little").hex(), state=t.state_bytes().hex()))
s0 = orc.random_fr(1, 11)
orc.lib.orc_transcript_append_scalar(C.byref(t), orc._p(s0))
ops.append(dict(op="append_scalar", arg=fr_hex(s0[0]), state=t.state_bytes().hex()))
sv = orc.random_fr(3, 12)
orc.lib.orc_transcript_append_scalars(C.byref(t), orc._p(sv), C.c_size_t(3))
ops.append(dict(op="append_scalars", arg="".join(fr_hex(x) for x in sv), state=t.state_bytes().hex()))
srs16 = orc.srs_powers(orc.random_fr(1, 0x51250001)[0], 17)
g = srs16[5]
buf = fq_canon(g["x"]).to_bytes(32, "big") + fq_canon(g["y"]).to_bytes(32, "big")
orc.lib.orc_transcript_append_bytes(C.byref(t), (C.c_uint8 * 64)(*buf), C.c_size_t(64))
ops.append(dict(op="append_point", arg=g1_hex(g), state=t.state_bytes().hex()))
raw = (C.c_uint64 * 2)(); orc.lib.orc_transcript_challenge_u128(C.byref(t), raw)
ops.append(dict(op="challenge_u128", out=u128_hex(raw[0] | (raw[1] << 64)), state=t.state_bytes().hex()))
s = orc.fr_array(1); orc.lib.orc_transcript_challenge_scalar(C.byref(t), orc._p(s))
ops.append(dict(op="challenge_scalar", out=fr_hex(s[0]), state=t.state_bytes().hex()))
r = orc.fr_array(1); orc.lib.orc_transcript_challenge_optimized(C.byref(t), raw, orc._p(r))
ops.append(dict(op="challenge_scalar_optimized", out_u128=u128_hex((raw[0] | (raw[1] << 64)) & ((1 << 125) - 1)), out=fr_hex(r[0]),
                state=t.state_bytes().hex()))
out["transcript"] = ops
# challenges
rng = np.random.default_rng(7)
vals = [0, 1, 2, 4, (1 << 128) - 1, (1 << 125) - 1] + [int.from_bytes(rng.bytes(16), "little") for _ in range(10)]
out["challenge_to_fr"] = [dict(u128=u128_hex(x), fr=fr_hex(orc.challenges_to_fr([x])[0])) for x in vals]
a = orc.random_fr(1, 13)[0]; cx = 0x0123456789abcdeffedcba9876543210
out["challenge_mul"] = dict(a=fr_hex(a), u128=u128_hex(cx), product=fr_hex(orc.fr_mul_arr(a, orc.challenges_to_fr([cx])[0])))
# sumcheck
n = 6
L, R = orc.random_fr(1 << n, 21), orc.random_fr(1 << n, 22)
claim = orc.dot_claim(L, R)
t = orc.new_transcript(b"synthetic_sc")
proof, ch, fin = orc.sumcheck_dot_prove(L, R, claim, t)
out["sumcheck"] = dict(n=n, left="".join(fr_hex(x) for x in L), right="".join(fr_hex(x) for x in R), claim=fr_hex(claim[0]),
                       compressed_polys=[(2).to_bytes(8, "little").hex() + fr_hex(proof[i][0]) + fr_hex(proof[i][1]) for i in range(n)],
                       challenges=[u128_hex(c & ((1 << 125) - 1)) for c in ch], final_left=fr_hex(fin[0]), final_right=fr_hex(fin[1]),
                       final_claim="", state=t.state_bytes().hex())
# hyperkzg
ell = 4
srs = srs16[:16]
pv = orc.random_fr(1 << ell, 31)
point = [int.from_bytes(rng.bytes(16), "little") & ((1 << 125) - 1) for _ in range(ell)]
com = orc.msm(srs, pv)
t = orc.new_transcript(b"TestEval")
c, w, v = orc.hyperkzg_open(srs, pv, point, t)
v = np.asarray(v).reshape(3, ell, 4)
pb = (ell - 1).to_bytes(8, "little").hex() + "".join(g1_hex(x) for x in c) + (3).to_bytes(8, "little").hex() + "".join(g1_hex(x) for x in w)
pb += (3).to_bytes(8, "little").hex() + "".join(ell.to_bytes(8, "little").hex() + "".join(fr_hex(v[i][j]) for j in range(ell)) for i in range(3))
out["hyperkzg"] = dict(ell=ell, g1_powers="".join(g1_hex(x) for x in srs), poly="".join(fr_hex(x) for x in pv), point=[u128_hex(x) for x in point],
                       eval="", commitment=g1_hex(com), proof=pb, state=t.state_bytes().hex())
json.dump(out, open(os.path.join(ROOT, "tests", "golden", "selfcheck_fixtures.json"), "w"), indent=0)
print("wrote tests/golden/selfcheck_fixtures.json")
