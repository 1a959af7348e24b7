"""Reference-HELD known-answer vectors for host-side pieces of the path (a20, a32): the small-integer cases of the reference's own unit
tests for UniPoly / gaussian_elimination and the doc test of interleave_bits (tests/golden/ref_unit_vectors.json, data only, extracted
by tools/extract_ref_unit_vectors.py), replayed through
  * the oracle (oracle/pymodel/poly.py and the C restatement behind oracle/orc.py),
  * the PRODUCT's host helpers (csrc/host_field.hpp, host_poly.hpp, compiled by g++ into tools/check_host_poly.cpp — no GPU needed),
  * and, for interleave_bits, the device kernel behind atlas_lookup_indices_from_operands (`-m gpu`)."""
import ctypes as C
import json
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
V = json.load(open(os.path.join(ROOT, "tests", "golden", "ref_unit_vectors.json")))


@pytest.fixture(scope="module")
def host_poly(tmp_path_factory):
    exe = tmp_path_factory.mktemp("hp") / "check_host_poly"
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", ROOT, os.path.join(ROOT, "tools", "check_host_poly.cpp"), "-o", str(exe)], check=True)

    def run(*args):
        out = subprocess.run([str(exe)] + [str(a) for a in args], capture_output=True, text=True, timeout=60)
        assert out.returncode == 0, out.stderr
        return [int(x) for x in out.stdout.split()]
    return run


def test_pymodel_unipoly_against_reference_vectors():
    from oracle.pymodel import poly as P
    for c in V["from_evals"]:
        got = P.unipoly_from_evals(c["evals"])
        assert got == c["coeffs"], c["test"]
        if "point" in c:
            assert P.unipoly_eval(got, c["point"]) == c["value"]
            hint = (c["evals"][0] + c["evals"][1]) % P.FR                       # compress / decompress with hint = e0 + e1
            comp = P.unipoly_compress(got)
            lin = (hint - 2 * comp[0] - sum(comp[1:])) % P.FR
            assert [comp[0], lin] + comp[1:] == got
    for c in V["from_coeff"]:
        assert P.unipoly_from_coeff(c["in"]) == c["out"], c["test"]
    t = V["toom"]["coeffs"]                                                       # evals at 0, 1, 2 and the leading coefficient
    assert P.unipoly_from_evals([P.unipoly_eval(t, x) for x in range(len(t))]) == t


def test_oracle_c_unipoly_against_reference_vectors():
    from oracle import orc
    for c in V["from_evals"]:
        if len(c["evals"]) not in (3, 4):
            continue
        e = c["evals"]
        hint = orc.from_ints([e[0] + e[1]])
        evals = orc.from_ints([e[0]] + e[2:])
        out = orc.fr_array(8)
        orc.lib.orc_unipoly_from_evals_and_hint.restype = C.c_size_t
        n = orc.lib.orc_unipoly_from_evals_and_hint(hint.ctypes.data_as(C.c_void_p), evals.ctypes.data_as(C.c_void_p), C.c_size_t(len(e) - 1),
                                                    out.ctypes.data_as(C.c_void_p))
        assert orc.to_ints(out[:n]) == c["coeffs"], c["test"]


def test_product_host_helpers_against_reference_vectors(host_poly):
    for c in V["from_evals"]:
        if len(c["evals"]) not in (3, 4):
            continue
        e = c["evals"]
        assert host_poly("from_evals_and_hint", len(e) - 1, e[0] + e[1], e[0], *e[2:]) == c["coeffs"], c["test"]
    for c in V["from_coeff"]:
        assert host_poly("trim", *c["in"]) == c["out"], c["test"]
    t = V["toom"]["coeffs"]
    ev = [sum(a * x ** k for k, a in enumerate(t)) for x in range(len(t) - 1)] + [t[-1]]
    assert host_poly("toom", *ev) == t
    g = V["gauss"]
    flat = [x for row in g["matrix"] for x in row]
    assert host_poly("gauss", len(g["matrix"]), *flat) == g["solution"]
    lq = V["linear_times_quadratic"]                                            # s(x) = (x + 1)(x^2 + 2x + 3): its values through the cubic path
    s = lq["coeffs"]
    val = lambda x: sum(a * x ** k for k, a in enumerate(s))
    assert val(0) + val(1) == lq["hint"]
    assert host_poly("from_evals_and_hint", 3, lq["hint"], val(0), val(2), val(3)) == s


def test_product_toom_interpolation_matches_the_matrix_form(host_poly):
    """from_evals_toom (UniPoly::from_evals_toom, unipoly.rs:103-134) runs in O(n) multiplications on the host's critical path (Newton
    differences + Horner over small integers); the n x n matrix form it replaced stays as the cross-check: 0 mismatches over lengths
    2 .. 40 x 20 inputs, and mul_small against the full multiplication on 200000 operands"""
    assert host_poly("toom_selfcheck") == [0]


def test_oracle_interleave_against_reference_doctest():
    from oracle import graph as OG
    from oracle import orc_ra
    for c in V["interleave_bits"]:
        assert orc_ra.interleave(c["even"], c["odd"]) == c["out"]
        assert int(OG.interleave_arr(np.array([c["even"]], dtype=np.int32), np.array([c["odd"]], dtype=np.int32))[0]) == c["out"]


@pytest.mark.gpu
def test_device_interleave_against_reference_doctest(atlas):
    from jolt_atlas_amd import instances as I
    for c in V["interleave_bits"]:
        left = atlas.TensorI32(np.full(64, c["even"], dtype=np.int32)); right = atlas.TensorI32(np.full(64, c["odd"], dtype=np.int32))
        dev = I.DeviceU64.from_operands(left, right)
        host = np.zeros(64, dtype=np.uint64)
        hip = C.CDLL("libamdhip64.so")
        assert hip.hipMemcpy(host.ctypes.data_as(C.c_void_p), dev.ptr, C.c_size_t(8 * 64), C.c_int(2)) == 0
        dev.free()
        assert (host == c["out"]).all()
