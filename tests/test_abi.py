"""CPU: the C-ABI library loads, exports every symbol include/atlas_hip.h declares, its
host-side entry points (transcript) are bit-exact, and device entry points fail loudly
without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "atlas_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(atlas_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported():
    import jolt_atlas_amd as A
    names = declared_symbols()
    assert len(names) >= 40
    missing = [n for n in names if not hasattr(A.lib, n)]
    assert not missing, missing


def test_product_does_not_link_the_oracle():
    import subprocess
    import jolt_atlas_amd as A
    out = subprocess.run(["ldd", A.LIB_PATH], capture_output=True, text=True).stdout
    assert "liboracle" not in out
    syms = subprocess.run(["nm", "-D", "--defined-only", A.LIB_PATH], capture_output=True, text=True).stdout
    assert " orc_" not in syms


def test_host_transcript_golden():
    import json
    import jolt_atlas_amd as A
    from oracle.pymodel import field as F
    d = json.load(open(os.path.join(ROOT, "tests", "golden", "transcript.json")))
    t = None
    for op, arg, state in d["ops"]:
        if op == "new":
            t = A.Blake2bTranscript(arg.encode())
        elif op == "append_message":
            t.append_message(arg.encode())
        elif op == "append_u64":
            t.append_u64(int(arg, 16))
        elif op == "append_scalar":
            t.append_scalar(np.array(F.limbs64(F.to_mont(int(arg, 16))), dtype=np.uint64))
        elif op == "append_scalars":
            t.append_scalars(np.array([F.limbs64(F.to_mont(int(x, 16))) for x in arg], dtype=np.uint64))
        elif op == "append_bytes":
            t.append_bytes(bytes.fromhex(arg))
        elif op == "challenge_u128":
            assert t.challenge_u128() == int(arg, 16)
        elif op == "challenge_scalar":
            assert F.from_mont(F.from_limbs64(t.challenge_scalar())) == int(arg, 16)
        elif op == "append_point":
            p = np.zeros(1, dtype=A.G1_DTYPE)
            if arg is None:
                p["infinity"] = 1
            else:
                p["x"][0] = F.limbs64(F.to_mont(int(arg[0], 16), F.FQ)); p["y"][0] = F.limbs64(F.to_mont(int(arg[1], 16), F.FQ))
            A._check(A.lib.atlas_transcript_append_point(C.byref(t.t), p.ctypes.data_as(C.c_void_p)))
        assert t.state.hex() == state, op
    assert t.n_rounds == d["n_rounds"]


def test_challenge_to_fr_modes():
    import json
    import jolt_atlas_amd as A
    from oracle.pymodel import field as F
    d = json.load(open(os.path.join(ROOT, "tests", "golden", "field.json")))
    try:
        for ch in d["challenges"]:
            A.set_challenge_mode(0)
            assert F.from_limbs64(A.challenge_to_fr(int(ch["c128"], 16))) == int(ch["mont_limbs_mode0"], 16)
            A.set_challenge_mode(1)
            assert F.from_limbs64(A.challenge_to_fr(int(ch["c128"], 16))) == int(ch["mont_limbs_mode1"], 16)
    finally:
        A.set_challenge_mode(0)
    with pytest.raises(A.AtlasError):
        A.set_challenge_mode(7)


def test_device_calls_fail_loudly_without_gpu():
    import jolt_atlas_amd as A
    if A.device_count() > 0:
        pytest.skip("a HIP device is present")
    with pytest.raises(A.AtlasError, match="no HIP device"):
        A.init(0)
    with pytest.raises(A.AtlasError):
        A.MultilinearPolynomial.from_fr(np.zeros((4, 4), dtype=np.uint64))
    with pytest.raises(A.AtlasError):
        A.SRS.generate(np.array([1, 0, 0, 0], dtype=np.uint64), 4)
