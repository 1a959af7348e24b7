"""TEST INFRASTRUCTURE ONLY: BN254 pairing over Python integers — the checker of the product's host pairing (csrc/host_pairing.hpp).

The pairing itself lives in arkworks (EXTERNAL to the reference tree; call sites joltworks/src/poly/commitment/hyperkzg/mod.rs:365
`P::multi_pairing([L, -R], [g2, beta_g2]).is_zero()`, kzg.rs:353-363).  HyperKZG::verify only asks whether a product of pairings is the
identity of G_T, so any bilinear non-degenerate pairing on (G1, G2) decides it identically; this is the ate pairing with Miller loop length
t - 1 = 6 x^2 (x = the BN parameter) on the sextic D-type twist, final exponentiation by plain square-and-multiply.  Tower:
Fq2 = Fq[u]/(u^2 + 1), Fq6 = Fq2[v]/(v^3 - xi), xi = 9 + u, Fq12 = Fq6[w]/(w^2 - v)."""

P = 0x30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47
R = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001
X = 4965661367192848881
ATE_LOOP = 6 * X * X                      # t - 1
assert P == 36 * X**4 + 36 * X**3 + 24 * X**2 + 6 * X + 1 and R == 36 * X**4 + 36 * X**3 + 18 * X**2 + 6 * X + 1
XI = (9, 1)
G2_GEN = ((10857046999023057135944570762232829481370756359578518086990519993285655852781,
           11559732032986387107991004021392285783925812861821192530917403151452391805634),
          (8495653923123431417604973247489272438418190587263600148770280649306958101930,
           4082367875863433681332203403145435568316851327593401208105741076214120093531))


# ---- Fq2
def f2_add(a, b): return ((a[0] + b[0]) % P, (a[1] + b[1]) % P)
def f2_sub(a, b): return ((a[0] - b[0]) % P, (a[1] - b[1]) % P)
def f2_neg(a): return (-a[0] % P, -a[1] % P)
def f2_mul(a, b): return ((a[0] * b[0] - a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)
def f2_scal(a, k): return (a[0] * k % P, a[1] * k % P)
def f2_inv(a):
    n = pow(a[0] * a[0] + a[1] * a[1], -1, P)
    return (a[0] * n % P, -a[1] * n % P)
F2_ZERO, F2_ONE = (0, 0), (1, 0)
def f2_mul_xi(a): return f2_mul(a, XI)


# ---- Fq6 = (c0, c1, c2) over Fq2, v^3 = xi
def f6_add(a, b): return tuple(f2_add(x, y) for x, y in zip(a, b))
def f6_sub(a, b): return tuple(f2_sub(x, y) for x, y in zip(a, b))
def f6_neg(a): return tuple(f2_neg(x) for x in a)
def f6_mul(a, b):
    a0, a1, a2 = a; b0, b1, b2 = b
    t0 = f2_add(f2_mul(a0, b0), f2_mul_xi(f2_add(f2_mul(a1, b2), f2_mul(a2, b1))))
    t1 = f2_add(f2_add(f2_mul(a0, b1), f2_mul(a1, b0)), f2_mul_xi(f2_mul(a2, b2)))
    t2 = f2_add(f2_add(f2_mul(a0, b2), f2_mul(a1, b1)), f2_mul(a2, b0))
    return (t0, t1, t2)
def f6_mul_v(a): return (f2_mul_xi(a[2]), a[0], a[1])
def f6_inv(a):
    a0, a1, a2 = a
    c0 = f2_sub(f2_mul(a0, a0), f2_mul_xi(f2_mul(a1, a2)))
    c1 = f2_sub(f2_mul_xi(f2_mul(a2, a2)), f2_mul(a0, a1))
    c2 = f2_sub(f2_mul(a1, a1), f2_mul(a0, a2))
    t = f2_add(f2_mul(a0, c0), f2_mul_xi(f2_add(f2_mul(a2, c1), f2_mul(a1, c2))))
    ti = f2_inv(t)
    return (f2_mul(c0, ti), f2_mul(c1, ti), f2_mul(c2, ti))
F6_ZERO, F6_ONE = (F2_ZERO,) * 3, (F2_ONE, F2_ZERO, F2_ZERO)


# ---- Fq12 = (c0, c1) over Fq6, w^2 = v
def f12_mul(a, b):
    t0, t1 = f6_mul(a[0], b[0]), f6_mul(a[1], b[1])
    return (f6_add(t0, f6_mul_v(t1)), f6_add(f6_mul(a[0], b[1]), f6_mul(a[1], b[0])))
def f12_conj(a): return (a[0], f6_neg(a[1]))
def f12_inv(a):
    t = f6_inv(f6_sub(f6_mul(a[0], a[0]), f6_mul_v(f6_mul(a[1], a[1]))))
    return (f6_mul(a[0], t), f6_neg(f6_mul(a[1], t)))
F12_ONE = (F6_ONE, F6_ZERO)
def f12_pow(a, e):
    r = F12_ONE
    for bit in bin(e)[2:]:
        r = f12_mul(r, r)
        if bit == "1":
            r = f12_mul(r, a)
    return r


# ---- the twist E'(Fq2): y^2 = x^3 + 3 / xi; affine points or None
B2 = f2_mul((3, 0), f2_inv(XI))
def g2_on_curve(Q):
    return Q is None or f2_sub(f2_mul(Q[1], Q[1]), f2_add(f2_mul(f2_mul(Q[0], Q[0]), Q[0]), B2)) == F2_ZERO
def g2_add(A, B):
    if A is None: return B
    if B is None: return A
    if A[0] == B[0]:
        if f2_add(A[1], B[1]) == F2_ZERO: return None
        lam = f2_mul(f2_scal(f2_mul(A[0], A[0]), 3), f2_inv(f2_scal(A[1], 2)))
    else:
        lam = f2_mul(f2_sub(B[1], A[1]), f2_inv(f2_sub(B[0], A[0])))
    x = f2_sub(f2_sub(f2_mul(lam, lam), A[0]), B[0])
    return (x, f2_sub(f2_mul(lam, f2_sub(A[0], x)), A[1]))
def g2_mul(Q, k):
    acc = None
    for bit in bin(k % R)[2:] if k % R else "":
        acc = g2_add(acc, acc)
        if bit == "1":
            acc = g2_add(acc, Q)
    return acc
def g2_neg(Q): return None if Q is None else (Q[0], f2_neg(Q[1]))


def _line(T, lam, Pt):
    """the line through T with (twist) slope lam at P = (xP, yP) in E(Fq): yP - lam xP w + (lam x_T - y_T) w^3, w^3 = v w"""
    xP, yP = Pt
    b = f2_neg(f2_scal(lam, xP)); c = f2_sub(f2_mul(lam, T[0]), T[1])
    return (((yP % P, 0), F2_ZERO, F2_ZERO), (b, c, F2_ZERO))


def miller(Pt, Q):
    """f_{t-1, Q}(P) for affine P in G1 (a pair of integers) and affine Q on the twist; either at infinity -> 1"""
    if Pt is None or Q is None:
        return F12_ONE
    f, T = F12_ONE, Q
    for bit in bin(ATE_LOOP)[3:]:
        lam = f2_mul(f2_scal(f2_mul(T[0], T[0]), 3), f2_inv(f2_scal(T[1], 2)))
        f = f12_mul(f12_mul(f, f), _line(T, lam, Pt))
        T = g2_add(T, T)
        if bit == "1":
            if T[0] == Q[0]:                               # T = -Q only at the very end of a loop over the group order; not reached for t - 1
                raise ValueError("degenerate addition in the Miller loop")
            lam = f2_mul(f2_sub(Q[1], T[1]), f2_inv(f2_sub(Q[0], T[0])))
            f = f12_mul(f, _line(T, lam, Pt))
            T = g2_add(T, Q)
    return f


HARD = (P**4 - P**2 + 1) // R
assert (P**4 - P**2 + 1) % R == 0
def final_exp(f):
    f = f12_mul(f12_conj(f), f12_inv(f))                  # ^(p^6 - 1)
    f = f12_mul(f12_pow(f, P * P), f)                     # ^(p^2 + 1)
    return f12_pow(f, HARD)


def multi_pairing_is_one(pairs):
    """prod e(P_i, Q_i) == 1 ?  pairs: (G1 affine (x, y) | None, twist affine | None)"""
    f = F12_ONE
    for Pt, Q in pairs:
        f = f12_mul(f, miller(Pt, Q))
    return final_exp(f) == F12_ONE
