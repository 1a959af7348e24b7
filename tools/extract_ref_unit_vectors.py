#!/usr/bin/env python3
"""The literal known-answer vectors the reference's own unit tests hold for host-side pieces of the path (small integers in Fr):

    joltworks/src/poly/unipoly.rs            test_from_evals_quad / _cubic / _edge_cases / _toom, test_from_linear_times_quadratic_with_hint,
                                             test_from_coeff_*, test_mul_unipoly_*                         (a20: UniPoly, CompressedUniPoly)
    joltworks/src/utils/gaussian_elimination.rs   test_gauss                                                 (a20: the Toom interpolation's solver)
    joltworks/src/utils/mod.rs               the doc test of interleave_bits                                (a32 / f1: lookup-index construction)

Data only (numbers parsed out of the test bodies, checked for presence so a change upstream is noticed):

    python tools/extract_ref_unit_vectors.py      # -> tests/golden/ref_unit_vectors.json"""
import json
import os
import re

REF = "/root/reference/joltworks/src"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "ref_unit_vectors.json")


def body(src, fn):
    i = src.index(f"fn {fn}")
    j = src.index("{", i)
    depth, k = 0, j
    while True:
        depth += {"{": 1, "}": -1}.get(src[k], 0)
        if depth == 0:
            return src[j:k + 1]
        k += 1


def ints(text):
    return [int(x) for x in re.findall(r"(?:from_u64\(|from\(|\b)(\d+)(?:u64)?(?:\.into\(\))?", text)]


def main():
    uni = open(f"{REF}/poly/unipoly.rs").read()
    out = {"source": "joltworks/src/poly/unipoly.rs, utils/gaussian_elimination.rs, utils/mod.rs (unit / doc tests; data only)",
           "generator": "tools/extract_ref_unit_vectors.py", "from_evals": [], "from_coeff": [], "mul": []}
    q = body(uni, "test_from_evals_quad_helper")
    e = [1] + [int(x) for x in re.findall(r"let e\d = F::from_u64\((\d+)u64\)", q)]           # e0 = F::one()
    c = [1 if "coeffs[0], F::one()" in q else None] + [int(x) for x in re.findall(r"coeffs\[[12]\], F::from_u64\((\d+)u64\)", q)]
    pt = int(re.search(r"evaluate::<F>\(&F::from_u64\((\d+)u64\)\)", q).group(1))
    out["from_evals"].append({"test": "test_from_evals_quad", "evals": e[:3], "coeffs": c, "point": pt, "value": e[3]})
    q = body(uni, "test_from_evals_cubic_helper")
    e = [1] + [int(x) for x in re.findall(r"let e\d = F::from_u64\((\d+)u64\)", q)]
    c = [1] + [int(x) for x in re.findall(r"coeffs\[[12]\], F::from_u64\((\d+)u64\)", q)] + [1 if "coeffs[3], F::one()" in q else None]
    pt = int(re.search(r"evaluate::<F>\(&F::from_u64\((\d+)u64\)\)", q).group(1))
    out["from_evals"].append({"test": "test_from_evals_cubic", "evals": e[:4], "coeffs": c, "point": pt, "value": e[4]})
    q = body(uni, "test_from_evals_edge_cases")
    m = re.search(r"let evals = vec!\[([^\]]*)\];\s*// evals of x\^2", q)
    ev = [int(x) for x in re.findall(r"(\d+)\.into\(\)", m.group(1))]
    m = re.search(r"coeffs: vec!\[([^\]]*)\]", q)
    out["from_evals"].append({"test": "test_from_evals_edge_cases (length 4 kept)", "evals": ev, "coeffs": [int(x) for x in re.findall(r"(\d+)\.into\(\)", m.group(1))]})
    assert "vec![42.into()]" in q and "Fr::zero()" in q
    out["from_evals"].append({"test": "test_from_evals_edge_cases (constant)", "evals": [42], "coeffs": [42]})
    out["from_evals"].append({"test": "test_from_evals_edge_cases (zero)", "evals": [0], "coeffs": [0]})
    q = body(uni, "test_from_evals_toom")
    m = re.search(r"from_coeff\(vec!\[([^\]]*)\]\)", q)
    out["toom"] = {"test": "test_from_evals_toom", "coeffs": [int(x) for x in re.findall(r"(\d+)\.into\(\)", m.group(1))]}
    q = body(uni, "test_from_linear_times_quadratic_with_hint")
    n = [int(x) for x in re.findall(r"Fr::from_u64\((\d+)u64\)", q)]
    out["linear_times_quadratic"] = {"test": "test_from_linear_times_quadratic_with_hint", "linear": n[0:2], "q0": n[2], "q2": n[3], "coeffs": n[4:8], "hint": n[8]}
    q = body(uni, "test_from_coeff_trims_leading_zeros")
    n = [int(x) for x in re.findall(r"Fr::from_u64\((\d+)u64\)", q)]
    out["from_coeff"].append({"test": "test_from_coeff_trims_leading_zeros", "in": n[0:2] + [0] * q.split("assert_eq!")[0].count("Fr::zero()"), "out": n[2:4]})
    q = body(uni, "test_from_coeff_all_zeros_is_zero_poly")
    out["from_coeff"].append({"test": "test_from_coeff_all_zeros_is_zero_poly", "in": [0] * q.split("assert!")[0].count("Fr::zero()"), "out": [0]})
    q = body(uni, "test_mul_unipoly_matches_expected_coefficients")
    n = [int(x) for x in re.findall(r"Fr::from_u64\((\d+)u64\)", q)]
    out["mul"].append({"test": "test_mul_unipoly_matches_expected_coefficients", "lhs": n[0:2], "rhs": n[2:4], "product": n[4:7]})
    q = body(uni, "test_mul_unipoly_normalizes_trailing_zeros")
    n = [int(x) for x in re.findall(r"Fr::from_u64\((\d+)u64\)", q)]
    out["mul"].append({"test": "test_mul_unipoly_normalizes_trailing_zeros", "lhs": [n[0], 0], "rhs": [n[1], 0, 0], "product": [n[2]]})
    g = body(open(f"{REF}/utils/gaussian_elimination.rs").read(), "test_gauss")
    rows = re.findall(r"vec!\[(Fr::[^\]]*)\]", g)
    conv = lambda s: [1 if t.strip() == "Fr::one()" else 0 if t.strip() == "Fr::zero()" else int(re.search(r"(\d+)u64", t).group(1)) for t in s.split(",") if t.strip()]
    out["gauss"] = {"test": "test_gauss", "matrix": [conv(r) for r in rows[:3]], "solution": conv(rows[3])}
    u = open(f"{REF}/utils/mod.rs").read()
    m = re.search(r"assert_eq!\(interleave_bits\(0b([01]+), 0b([01]+)\), 0b([01]+)\)", u)
    out["interleave_bits"] = [{"even": int(m.group(1), 2), "odd": int(m.group(2), 2), "out": int(m.group(3), 2)}]
    with open(OUT, "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")
    print(json.dumps(out)[:1500])


if __name__ == "__main__":
    main()
