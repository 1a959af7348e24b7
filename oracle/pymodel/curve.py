"""BN254 G1 with plain Python integers (affine, textbook formulas).

TEST INFRASTRUCTURE ONLY (oracle).  Independent of oracle/curve.c; pins it and generates
the golden MSM / HyperKZG fixtures.  Curve: y^2 = x^3 + 3 over Fq, generator (1, 2),
group order = Fr modulus (public BN254 parameters)."""
from .field import FQ, FR

G = (1, 2)
INF = None


def on_curve(P):
    return P is None or (P[1] * P[1] - P[0] ** 3 - 3) % FQ == 0


def add(P, Q):
    if P is None:
        return Q
    if Q is None:
        return P
    x1, y1 = P
    x2, y2 = Q
    if x1 == x2:
        if (y1 + y2) % FQ == 0:
            return None
        lam = 3 * x1 * x1 * pow(2 * y1, -1, FQ) % FQ
    else:
        lam = (y2 - y1) * pow(x2 - x1, -1, FQ) % FQ
    x3 = (lam * lam - x1 - x2) % FQ
    return (x3, (lam * (x1 - x3) - y1) % FQ)


def neg(P):
    return None if P is None else (P[0], (-P[1]) % FQ)


def mul(P, k):
    k %= FR
    acc = None
    while k:
        if k & 1:
            acc = add(acc, P)
        P = add(P, P)
        k >>= 1
    return acc


def msm(bases, scalars):
    acc = None
    for b, s in zip(bases, scalars):
        acc = add(acc, mul(b, s))
    return acc


def srs_powers(tau, n):
    """bases[i] = tau^(i+1) * G (reference layout, hyperkzg/kzg.rs:47-53)."""
    out, pw = [], tau % FR
    for _ in range(n):
        out.append(mul(G, pw))
        pw = pw * tau % FR
    return out
