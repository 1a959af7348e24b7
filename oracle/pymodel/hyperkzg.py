"""HyperKZG open / verify with plain Python integers.

TEST INFRASTRUCTURE ONLY (oracle).  Restates
  HyperKZG::open            joltworks/src/poly/commitment/hyperkzg/mod.rs:400-447
  kzg_open_batch            hyperkzg/mod.rs:231-280
  compute_witness_polynomial hyperkzg/mod.rs:213-229
  verify_inner / kzg_verify_batch  hyperkzg/mod.rs:451-509, 283-366
The pairing check e(L, H) == e(R, tau H) is replaced by the equivalent G1 statement
L == tau * R, which needs the SRS trapdoor — available because the build's test SRS is
generated from a known tau (bases[i] = tau^(i+1) G)."""
from . import curve as EC
from .field import FR, challenge_to_fr


def commit(srs, coeffs):
    return EC.msm(srs[:len(coeffs)], coeffs)


def eval_univariate(f, x):
    acc, pw = 0, 1
    for c in f:
        acc = (acc + c * pw) % FR
        pw = pw * x % FR
    return acc


def witness_poly(f, u):
    d = len(f)
    h = [0] * d
    for i in range(d - 1, 0, -1):
        h[i - 1] = (f[i] + h[i] * u) % FR
    return h


def open_(srs, poly, point_c128, t):
    """point_c128: raw u128 challenges (MontU128Challenge), point[0] = MSB variable."""
    ell = len(point_c128)
    point = [challenge_to_fr(c) for c in point_c128]
    assert len(poly) == 1 << ell
    polys = [list(poly)]
    for i in range(ell - 1):
        p = polys[i]
        r = point[ell - i - 1]
        polys.append([(r * (p[2 * j + 1] - p[2 * j]) + p[2 * j]) % FR for j in range(len(p) // 2)])
    com = [commit(srs, p) for p in polys[1:]]
    t.append_points(com)
    r = t.challenge_scalar()
    u = [r, (-r) % FR, r * r % FR]
    v = [[eval_univariate(f, ui) for f in polys] for ui in u]
    t.append_scalars([x for row in v for x in row])
    q = t.challenge_scalar_powers(ell)
    n = len(poly)
    B = [sum(q[j] * polys[j][i] for j in range(ell) if i < len(polys[j])) % FR for i in range(n)]
    w = [commit(srs, witness_poly(B, ui)) for ui in u]
    t.append_points(w)
    t.challenge_scalar()
    return com, w, v


def verify_trapdoor(srs, tau, C, point_c128, y, proof, t):
    com, w, v = proof
    ell = len(point_c128)
    point = [challenge_to_fr(c) for c in point_c128]
    t.append_points(com)
    r = t.challenge_scalar()
    if r == 0 or C is None:
        return False
    com = [C] + list(com)
    u = [r, (-r) % FR, r * r % FR]
    if len(v) != 3 or any(len(row) != ell for row in v):
        return False
    ypos, yneg, Y = v[0], v[1], list(v[2]) + [y % FR]
    for i in range(ell):
        lhs = 2 * r * Y[i + 1] % FR
        x = point[ell - i - 1]
        rhs = (r * (1 - x) * (ypos[i] + yneg[i]) + x * (ypos[i] - yneg[i])) % FR
        if lhs != rhs:
            return False
    # kzg_verify_batch
    t.append_scalars([x for row in v for x in row])
    q = t.challenge_scalar_powers(ell)
    t.append_points(w)
    d0 = t.challenge_scalar()
    d1 = d0 * d0 % FR
    mult = (1 + d0 + d1) % FR
    Bu = [sum(a * b for a, b in zip(row, q)) % FR for row in v]
    g1 = srs[0]
    bases = com + [w[0], w[1], w[2], g1]
    scal = [qj * mult % FR for qj in q] + [u[0], u[1] * d0 % FR, u[2] * d1 % FR,
                                          (-(Bu[0] + d0 * Bu[1] + d1 * Bu[2])) % FR]
    L = EC.msm(bases, scal)
    R = EC.add(EC.add(w[0], EC.mul(w[1], d0)), EC.mul(w[2], d1))
    return L == EC.mul(R, tau)
