"""Multilinear / univariate polynomial helpers over Fr (plain ints mod r).

TEST INFRASTRUCTURE ONLY (oracle).  Restates, with /root/reference line refs:
  eq_evals            joltworks/src/poly/eq_poly.rs:149-167 (big-endian table)
  bind                joltworks/src/poly/dense_mlpoly.rs:91-101,209-239
  sumcheck_evals      joltworks/src/poly/multilinear_polynomial.rs:873-905
  UniPoly             joltworks/src/poly/unipoly.rs:55-133,219-245,307-318,502-558
"""
from .field import FR

HIGH_TO_LOW = 0
LOW_TO_HIGH = 1


def eq_evals(r, scaling=1):
    """EqPolynomial::evals — r[0] is the MSB of the index (eq_poly.rs:63-76)."""
    evals = [scaling % FR]
    for rj in r:
        nxt = [0] * (2 * len(evals))
        for i, s in enumerate(evals):
            hi = s * rj % FR
            nxt[2 * i + 1] = hi
            nxt[2 * i] = (s - hi) % FR
        evals = nxt
    return evals


def eq_mle(x, y):
    """EqPolynomial::mle (eq_poly.rs:23-36)."""
    acc = 1
    for a, b in zip(x, y):
        acc = acc * ((a * b + (1 - a) * (1 - b)) % FR) % FR
    return acc


def bind(z, r, order):
    """DensePolynomial::bind: Z'[i] = a + r*(b-a)."""
    n = len(z) // 2
    if order == HIGH_TO_LOW:
        return [(z[i] + r * (z[i + n] - z[i])) % FR for i in range(n)]
    return [(z[2 * i] + r * (z[2 * i + 1] - z[2 * i])) % FR for i in range(n)]


def sumcheck_evals(z, i, degree, order):
    """Evaluations of the 1-variable restriction at 0, 2, 3, ..., degree."""
    n = len(z) // 2
    if order == HIGH_TO_LOW:
        a, b = z[i], z[i + n]
    else:
        a, b = z[2 * i], z[2 * i + 1]
    m = (b - a) % FR
    out = [a]
    e = b
    for _ in range(1, degree):
        e = (e + m) % FR
        out.append(e)
    return out


def evaluate(z, r):
    """MLE evaluation, r[0] = MSB (multilinear_polynomial.rs:766-862)."""
    for rj in r:
        z = bind(z, rj, HIGH_TO_LOW)
    return z[0]


# ---------------------------------------------------------------- UniPoly

def _inv(a):
    return pow(a, -1, FR)


def unipoly_from_evals(evals):
    """UniPoly::from_evals (unipoly.rs:55-89): evals at 0..n-1; fixed length 3/4
    for degree 2/3 (no trimming), Vandermonde + trim otherwise."""
    n = len(evals)
    e = [x % FR for x in evals]
    if n == 3:
        c0 = e[0]
        c2 = (e[0] - 2 * e[1] + e[2]) * _inv(2) % FR
        c1 = (e[1] - e[0] - c2) % FR
        return [c0, c1, c2]
    if n == 4:
        c0 = e[0]
        c3 = (e[3] - e[0] + 3 * (e[1] - e[2])) * _inv(6) % FR
        c2 = ((e[0] - 2 * e[1] + e[2]) * _inv(2) - 3 * c3) % FR
        c1 = (e[1] - e[0] - c2 - c3) % FR
        return [c0, c1, c2, c3]
    # generic Lagrange interpolation on 0..n-1, then from_coeff trimming
    coeffs = [0] * n
    for i in range(n):
        num = [1]
        den = 1
        for j in range(n):
            if j == i:
                continue
            num = [(a - j * b) % FR for a, b in zip([0] + num, num + [0])]
            den = den * (i - j) % FR
        s = e[i] * _inv(den) % FR
        for k in range(n):
            coeffs[k] = (coeffs[k] + s * num[k]) % FR
    return unipoly_from_coeff(coeffs)


def unipoly_from_coeff(c):
    c = [x % FR for x in c]
    while c and c[-1] == 0:
        c.pop()
    return c or [0]


def unipoly_from_evals_and_hint(hint, evals):
    """unipoly.rs:91-98 : evals at 0,2,3,.. plus hint = p(0)+p(1)."""
    e = list(evals)
    e.insert(1, (hint - e[0]) % FR)
    return unipoly_from_evals(e)


def unipoly_eval(c, x):
    acc = 0
    for a in reversed(c):
        acc = (acc * x + a) % FR
    return acc


def unipoly_compress(c):
    """unipoly.rs:307-318 : drop the linear term."""
    if len(c) < 2:
        return list(c)
    return [c[0]] + list(c[2:])


def compressed_eval_from_hint(cc, hint, x):
    """CompressedUniPoly::eval_from_hint (unipoly.rs:519-533)."""
    lin = (hint - 2 * cc[0] - sum(cc[1:])) % FR
    acc = (cc[0] + x * lin) % FR
    p = x
    for a in cc[1:]:
        p = p * x % FR
        acc = (acc + a * p) % FR
    return acc


def append_compressed(t, cc):
    """AppendToTranscript for CompressedUniPoly (unipoly.rs:550-558)."""
    t.append_message(b"UniPoly_begin")
    for a in cc:
        t.append_scalar(a)
    t.append_message(b"UniPoly_end")
