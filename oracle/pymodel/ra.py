"""Naive full-table models of the one-hot "ra" sumcheck instances (second, independent restatement
used to pin oracle/ra.c).  TEST INFRASTRUCTURE ONLY (oracle).

Each model keeps the whole summand as dense tables and produces the round polynomial by
evaluating it at 0..deg and interpolating — no Gruen split, no prefix tables, no grid tricks — so
agreement with oracle/ra.c checks those optimisations as restated from
  subprotocols/ra_virtual.rs:97-145, mles_product_sum.rs:15-131,330-376,
  subprotocols/booleanity.rs:169-348, subprotocols/hamming_weight.rs:106-148,
  poly/split_eq_poly.rs, poly/ra_poly.rs, utils/expanding_table.rs.
Indices: list of ints, negative = None.
"""
from .field import FR
from . import poly as P

L2H = P.LOW_TO_HIGH


def interpolate(evals):
    """coefficients of the polynomial through (0, e0), (1, e1), ... (exact, mod FR)."""
    n = len(evals)
    coeffs = [0] * n
    for i, yi in enumerate(evals):
        num = [1]
        den = 1
        for j in range(n):
            if j == i:
                continue
            num = [(a - j * b) % FR for a, b in zip([0] + num, num + [0])]
            den = den * (i - j) % FR
        s = yi * pow(den, -1, FR) % FR
        for k in range(n):
            coeffs[k] = (coeffs[k] + s * num[k]) % FR
    return coeffs


def from_coeff(c):
    c = list(c)
    while c and c[-1] == 0:
        c.pop()
    return c or [0]


def _ext(z, i, npts):
    """values at X = 0..npts-1 of the LowToHigh restriction of table z at pair i."""
    a, b = z[2 * i], z[2 * i + 1]
    m = (b - a) % FR
    return [(a + x * m) % FR for x in range(npts)]


class RaVirtualModel:
    """sum_j eq(r_cycle, j) prod_i ra_i(j), ra_i(j) = eq(chunk_i, idx_i[j])."""

    def __init__(self, H_indices, r_address_chunks, r_cycle):
        self.d = len(H_indices)
        self.ra = []
        for idx, chunk in zip(H_indices, r_address_chunks):
            F = P.eq_evals(chunk)
            self.ra.append([0 if k < 0 else F[k] for k in idx])
        self.eq = P.eq_evals(r_cycle)
        self._n = len(r_cycle)

    def num_rounds(self):
        return self._n

    def input_claim(self):
        acc = 0
        for j in range(len(self.eq)):
            t = self.eq[j]
            for ra in self.ra:
                t = t * ra[j] % FR
            acc = (acc + t) % FR
        return acc

    def compute_message(self, rnd, previous_claim):
        npts = self.d + 2
        ev = [0] * npts
        for i in range(len(self.eq) // 2):
            cols = [_ext(self.eq, i, npts)] + [_ext(ra, i, npts) for ra in self.ra]
            for x in range(npts):
                t = 1
                for c in cols:
                    t = t * c[x] % FR
                ev[x] = (ev[x] + t) % FR
        assert (ev[0] + ev[1]) % FR == previous_claim % FR
        return from_coeff(interpolate(ev))

    def ingest_challenge(self, r, rnd):
        self.eq = P.bind(self.eq, r, L2H)
        self.ra = [P.bind(ra, r, L2H) for ra in self.ra]

    def finals(self):
        return [ra[0] for ra in self.ra]


class BooleanityModel:
    """sum_{k,j} eq(r_address,k) eq(r_cycle,j) sum_i gamma_i (ra_i(k,j)^2 - ra_i(k,j)); address
    variables first (LowToHigh), then cycle variables (LowToHigh)."""

    def __init__(self, H_indices, log_k, gammas, r_address, r_cycle):
        K = 1 << log_k
        T = len(H_indices[0])
        self.log_k = log_k
        ea, ec = P.eq_evals(r_address), P.eq_evals(r_cycle)
        self.eq = [ec[j] * ea[k] % FR for j in range(T) for k in range(K)]
        self.ra = [[1 if idx[j] == k else 0 for j in range(T) for k in range(K)] for idx in H_indices]
        self.gammas = list(gammas)
        self._n = log_k + len(r_cycle)

    def num_rounds(self):
        return self._n

    def compute_message(self, rnd, previous_claim):
        ev = [0] * 4
        for i in range(len(self.eq) // 2):
            e = _ext(self.eq, i, 4)
            cols = [_ext(ra, i, 4) for ra in self.ra]
            for x in range(4):
                s = 0
                for g, c in zip(self.gammas, cols):
                    s = (s + g * (c[x] * c[x] - c[x])) % FR
                ev[x] = (ev[x] + e[x] * s) % FR
        assert (ev[0] + ev[1]) % FR == previous_claim % FR
        c = interpolate(ev)
        return c if rnd < self.log_k else from_coeff(c)     # phase 2 scales by eq_r_r -> from_coeff

    def ingest_challenge(self, r, rnd):
        self.eq = P.bind(self.eq, r, L2H)
        self.ra = [P.bind(ra, r, L2H) for ra in self.ra]

    def finals(self):
        return [ra[0] for ra in self.ra]


class HammingModel:
    """sum_k sum_i gamma_i G_i[k], LowToHigh, degree 1."""

    def __init__(self, G, gamma_powers):
        self.ra = [list(g) for g in G]
        self.g = list(gamma_powers)
        self._n = len(G[0]).bit_length() - 1

    def num_rounds(self):
        return self._n

    def compute_message(self, rnd, previous_claim):
        ev = [0, 0]
        for ra, g in zip(self.ra, self.g):
            for i in range(len(ra) // 2):
                ev[0] = (ev[0] + g * ra[2 * i]) % FR
                ev[1] = (ev[1] + g * ra[2 * i + 1]) % FR
        assert (ev[0] + ev[1]) % FR == previous_claim % FR
        return from_coeff([ev[0], (ev[1] - ev[0]) % FR])

    def ingest_challenge(self, r, rnd):
        self.ra = [P.bind(ra, r, L2H) for ra in self.ra]

    def finals(self):
        return [ra[0] for ra in self.ra]


H2L = P.HIGH_TO_LOW


def _ext_h(z, i, npts):
    """values at X = 0..npts-1 of the HighToLow restriction of table z at index i."""
    n = len(z) // 2
    a, b = z[i], z[i + n]
    m = (b - a) % FR
    return [(a + x * m) % FR for x in range(npts)]


class DenseOpeningModel:
    """sum_j eq(point, j) P(j), HighToLow, degree 2 (opening_reduction.rs:355-425), naive."""

    def __init__(self, poly, point):
        self.P = list(poly)
        self.eq = P.eq_evals(point)
        self._n = len(point)

    def num_rounds(self):
        return self._n

    def compute_message(self, rnd, previous_claim):
        ev = [0, 0, 0]
        for i in range(len(self.P) // 2):
            e, p = _ext_h(self.eq, i, 3), _ext_h(self.P, i, 3)
            for x in range(3):
                ev[x] = (ev[x] + e[x] * p[x]) % FR
        assert (ev[0] + ev[1]) % FR == previous_claim % FR
        return interpolate(ev)                       # UniPoly::from_evals of 3 points: fixed length

    def ingest_challenge(self, r, rnd):
        self.P = P.bind(self.P, r, H2L)
        self.eq = P.bind(self.eq, r, H2L)

    def finals(self):
        return [self.P[0]]


class OneHotOpeningModel:
    """sum_{k,j} eq(r_address,k) eq(r_cycle,j) [idx_j = k]; address variables first, everything
    HighToLow (opening_reduction.rs:532-723), naive over the K*T table (index k*T + j)."""

    def __init__(self, idx, log_K, r_address, r_cycle):
        K, T = 1 << log_K, len(idx)
        self.log_K = log_K
        ea, ec = P.eq_evals(r_address), P.eq_evals(r_cycle)
        self.eq = [ea[k] * ec[j] % FR for k in range(K) for j in range(T)]
        self.ra = [1 if idx[j] == k else 0 for k in range(K) for j in range(T)]
        self._n = log_K + len(r_cycle)

    def num_rounds(self):
        return self._n

    def input_claim(self):
        return sum(e * a for e, a in zip(self.eq, self.ra)) % FR

    def compute_message(self, rnd, previous_claim):
        ev = [0, 0, 0]
        for i in range(len(self.eq) // 2):
            e, p = _ext_h(self.eq, i, 3), _ext_h(self.ra, i, 3)
            for x in range(3):
                ev[x] = (ev[x] + e[x] * p[x]) % FR
        assert (ev[0] + ev[1]) % FR == previous_claim % FR
        c = interpolate(ev)
        return c if rnd < self.log_K else from_coeff(c)

    def ingest_challenge(self, r, rnd):
        self.eq = P.bind(self.eq, r, H2L)
        self.ra = P.bind(self.ra, r, H2L)

    def finals(self):
        return [self.ra[0]]


class PsReluModel:
    """Read-raf sumcheck of the ReLU lookup, unary: sum_{k,t} eq(r_node,t) [k = idx_t] (Val(k) + gamma SId(k)),
    address variables MSB first, then cycle variables LowToHigh (ps_shout/mod.rs, unary.rs).  Closed-form
    model: no prefix/suffix tables, no checkpoints — each lookup's contribution is evaluated from the
    multilinear extensions Val~(x) = (1 - x_0) sum_{i>=1} x_i 2^(N-1-i) (relu.rs:30-43) and
    SId~(x) = sum_i x_i 2^(N-1-i) - x_0 2^N (signed_identity_poly.rs:44-60)."""

    def __init__(self, idx, N, r_node, gamma):
        self.idx, self.N, self.gamma = list(idx), N, gamma
        self.u = P.eq_evals(r_node)
        self.r_addr = []
        self._n = N + len(r_node)
        self.eq = None

    def num_rounds(self):
        return self._n

    def _W(self, x):
        N = self.N
        word = sum(x[i] * (1 << (N - 1 - i)) for i in range(1, N)) % FR
        val = (1 - x[0]) * word % FR
        sid = (sum(x[i] * (1 << (N - 1 - i)) for i in range(N)) - x[0] * (1 << N)) % FR
        return (val + self.gamma * sid) % FR

    def input_claim(self):
        N = self.N
        acc = 0
        for t, k in enumerate(self.idx):
            bits = [(k >> (N - 1 - i)) & 1 for i in range(N)]
            acc = (acc + self.u[t] * self._W(bits)) % FR
        return acc

    def compute_message(self, rnd, previous_claim):
        N = self.N
        ev = [0, 0, 0]
        if rnd < N:
            j = rnd
            for t, k in enumerate(self.idx):
                bits = [(k >> (N - 1 - i)) & 1 for i in range(N)]
                w = self.u[t]
                for i in range(j):
                    w = w * (self.r_addr[i] if bits[i] else (1 - self.r_addr[i])) % FR
                for X in range(3):
                    e = X if bits[j] else (1 - X)
                    x = self.r_addr[:j] + [X] + bits[j + 1:]
                    ev[X] = (ev[X] + w * e * self._W(x)) % FR
        else:
            for i in range(len(self.eq) // 2):
                e, a = _ext(self.eq, i, 3), _ext(self.ra, i, 3)
                for X in range(3):
                    ev[X] = (ev[X] + e[X] * a[X] % FR * self.wv) % FR
        assert (ev[0] + ev[1]) % FR == previous_claim % FR
        return interpolate(ev)

    def ingest_challenge(self, r, rnd):
        N = self.N
        if rnd < N:
            self.r_addr.append(r)
            if rnd == N - 1:
                self.ra = []
                for k in self.idx:
                    bits = [(k >> (N - 1 - i)) & 1 for i in range(N)]
                    w = 1
                    for i in range(N):
                        w = w * (self.r_addr[i] if bits[i] else (1 - self.r_addr[i])) % FR
                    self.ra.append(w)
                self.eq = list(self.u)
                self.wv = self._W(self.r_addr)
        else:
            self.eq = P.bind(self.eq, r, L2H)
            self.ra = P.bind(self.ra, r, L2H)

    def finals(self):
        return [self.ra[0]]


class PsRightShiftModel(PsReluModel):
    """The same read-raf sumcheck over RightShiftTable<N> (lookup_tables/right_shift.rs:17-43): Val~(x) =
    sum_{i <= N-1-D} x_i 2^(N-1-i-D), the multilinear extension of k >> D; RAF = SignedIdentity."""

    def __init__(self, idx, N, shift, r_node, gamma):
        super().__init__(idx, N, r_node, gamma)
        self.shift = shift

    def _W(self, x):
        N, D = self.N, self.shift
        val = sum(x[i] * (1 << (N - 1 - i - D)) for i in range(N - D)) % FR
        sid = (sum(x[i] * (1 << (N - 1 - i)) for i in range(N)) - x[0] * (1 << N)) % FR
        return (val + self.gamma * sid) % FR


class PsIdentityModel(PsReluModel):
    """IdentityRC: sum_{k,t} eq(r_node,t) [k = idx_t] Id(k), Id~(x) = sum_i x_i 2^(L-1-i)
    (identity_range_check.rs:196-420), same closed-form evaluation as PsReluModel."""

    def __init__(self, idx, log_K, r_node):
        super().__init__(idx, log_K, r_node, 0)

    def _W(self, x):
        N = self.N
        return sum(x[i] * (1 << (N - 1 - i)) for i in range(N)) % FR


class PsClampModel(PsReluModel):
    """ClampBoundedTable<N, bound, symmetric> read-raf: Val~ as ClampBoundedTable::evaluate_mle
    (lookup_tables/clamp.rs:140-195) + gamma * SignedIdentity."""

    def __init__(self, idx, N, bound, symmetric, r_node, gamma):
        super().__init__(idx, N, r_node, gamma)
        self.bound, self.symmetric = bound, symmetric

    def _W(self, x):
        N, B = self.N, self.bound
        h = N - B
        haz = hao = 1
        for i in range(h):
            haz = haz * (1 - x[i]) % FR
            hao = hao * x[i] % FR
        if not self.symmetric:
            hao = 0
        lw = sum(x[i] * (1 << (N - 1 - i)) for i in range(h, N)) % FR
        U = (1 << B) - 1
        LC = 2 * U + 1 if self.symmetric else U
        val = (U - x[0] * LC + haz * (lw - U) + hao * lw) % FR
        sid = (sum(x[i] * (1 << (N - 1 - i)) for i in range(N)) - x[0] * (1 << N)) % FR
        return (val + self.gamma * sid) % FR


class PsUltModel(PsReluModel):
    """Binary read-raf with UnsignedLessThanTable<32>: W = LT~ + gamma SignedLeft~ + gamma^2 SignedRight~ over the
    64 interleaved variables (unsigned_less_than.rs:27-43, signed_identity_poly.rs:226-244)."""

    def __init__(self, idx, r_node, gamma):
        super().__init__(idx, 64, r_node, gamma)

    def _W(self, v):
        lt, eq, lo, ro = 0, 1, 0, 0
        for i in range(32):
            x, y = v[2 * i], v[2 * i + 1]
            lt = (lt + (1 - x) * y * eq) % FR
            eq = eq * (x * y + (1 - x) * (1 - y)) % FR
            lo = (lo + x * (1 << (31 - i))) % FR
            ro = (ro + y * (1 << (31 - i))) % FR
        lo = (lo - v[0] * (1 << 32)) % FR
        ro = (ro - v[1] * (1 << 32)) % FR
        return (lt + self.gamma * lo + self.gamma * self.gamma * ro) % FR


class ElementwiseModel:
    """sum_x eq(r_node_output, x) f(operands(x)) for the element-wise operators (ops/add.rs, sub.rs, neg.rs,
    square.rs, mul.rs, iff.rs, cube.rs): dense tables, round polynomial by evaluation at 0..deg and
    interpolation.  Add / Sub / Neg answer with 3 coefficients, Square / Mul / Iff with 4 (UniPoly::from_evals
    keeps the length), Cube goes through from_coeff (trimmed) like every mles_product_sum user."""

    F = {
        "add": (2, lambda v: (v[0] + v[1]) % FR),
        "sub": (2, lambda v: (v[0] - v[1]) % FR),
        "neg": (2, lambda v: (-v[0]) % FR),
        "square": (3, lambda v: v[0] * v[0] % FR),
        "mul": (3, lambda v: v[0] * v[1] % FR),
        "iff": (3, lambda v: (v[0] * v[1] + (1 - v[0]) * v[2]) % FR),
        "cube": (4, lambda v: v[0] * v[0] % FR * v[0] % FR),
    }

    def __init__(self, op, operands, r_node_output, constants=()):
        self.op = op
        if op == "div":            # operands left, right, q, R: right * q + R - left   (ops/div.rs:329-351)
            self.deg, self.f = 3, lambda v: (v[1] * v[2] + v[3] - v[0]) % FR
        elif op == "rsqrt":        # input, quotient, output, div_rem, sqrt_rem; constants S^3, gamma (ops/rsqrt.rs:390-421)
            s3, gam = constants
            self.deg, self.f = 3, lambda v: (v[0] * v[1] + v[3] - s3 + gam * (v[2] * v[2] + v[4] - v[1])) % FR
        elif op == "hamming_bool":  # hw_0 .. hw_{d-1}; constants gamma_powers (subprotocols/hamming_booleanity.rs:131-156)
            gp = list(constants)
            self.deg, self.f = 3, lambda v: sum(g_ * (h * h - h) for g_, h in zip(gp, v)) % FR
        elif op == "teleport_div":  # input, quotient, remainder; constant tau (neural_teleport/division.rs:231-249)
            tau = constants[0]
            self.deg, self.f = 2, lambda v: (tau * v[1] + v[2] - v[0]) % FR
        elif op == "dot":          # pairs (input_t, selector_t), no eq factor (ops/reshape.rs, slice.rs, concat.rs)
            self.deg, self.f = 2, lambda v: sum(v[i] * v[i + 1] for i in range(0, len(v), 2)) % FR
        elif op == "gather":       # ra, dictionary, identity; constant gamma (ops/gather/mod.rs:232-261)
            gam = constants[0]
            self.deg, self.f = 2, lambda v: v[0] * (v[1] + gam * v[2]) % FR
        else:
            self.deg, self.f = self.F[op]
        self.x = [list(o) for o in operands]
        self.eq = P.eq_evals(r_node_output) if op not in ("dot", "gather") else [1] * len(self.x[0])
        self.no_eq = op in ("dot", "gather")
        self._n = len(r_node_output)

    def num_rounds(self):
        return self._n

    def input_claim(self):
        return sum(self.eq[j] * self.f([x[j] for x in self.x]) for j in range(len(self.eq))) % FR

    def compute_message(self, rnd, previous_claim):
        npts = self.deg + 1
        ev = [0] * npts
        for i in range(len(self.eq) // 2):
            e = _ext(self.eq, i, npts)
            cols = [_ext(x, i, npts) for x in self.x]
            for X in range(npts):
                ev[X] = (ev[X] + e[X] * self.f([c[X] for c in cols])) % FR
        assert (ev[0] + ev[1]) % FR == previous_claim % FR
        c = interpolate(ev)
        return from_coeff(c) if self.op == "cube" else c

    def ingest_challenge(self, r, rnd):
        self.eq = [1] * (len(self.eq) // 2) if self.no_eq else P.bind(self.eq, r, L2H)
        self.x = [P.bind(x, r, L2H) for x in self.x]

    def finals(self):
        return [x[0] for x in self.x]


class SoftmaxModel:
    """Dense-table models of the softmax_last_axis provers (ops/softmax_last_axis/exp_sum.rs, max.rs, recip_mult.rs) and
    of SumAxisProver (ops/sum/axis.rs): every factor — including the eq factor that the provers keep as a K-entry
    table or a Gruen split — is a full table over (k, j); the round polynomial comes from evaluation and
    interpolation, with the coefficient-vector lengths the reference produces (from_coeff trims the degree-1
    messages; from_evals of 3 / 4 points keeps its length)."""

    def __init__(self, kind, a, b, log_K, log_N, r):
        self.kind, self.log_K, self.log_N = kind, log_K, log_N
        N = 1 << log_N
        self.tabs = [list(a)]
        if kind == "max":
            self.tabs.append(list(b))
        if kind == "recip":
            self.tabs.append([b[kj >> log_N] for kj in range(len(a))])
        if kind == "mean_sq":      # MeanOfSquaresReductionProver (ops/mean_of_squares.rs:363-398): operand^2 eq(retained), HighToLow
            self.tabs.append(list(a))
        if kind in ("exp_sum", "max", "mean_sq"):
            ek = P.eq_evals(r)
            self.tabs.append([ek[kj >> log_N] for kj in range(len(a))])
        if kind == "recip":
            self.tabs.append(P.eq_evals(r))
        self.order = P.HIGH_TO_LOW if kind in ("sum_axis", "mean_sq") else L2H
        self._n = log_K + log_N

    def num_rounds(self):
        return self._n

    def input_claim(self):
        acc = 0
        for i in range(len(self.tabs[0])):
            t = 1
            for tab in self.tabs:
                t = t * tab[i] % FR
            acc = (acc + t) % FR
        return acc

    def _ext(self, z, i, npts):
        half = len(z) // 2
        a, b = (z[2 * i], z[2 * i + 1]) if self.order == L2H else (z[i], z[i + half])
        m = (b - a) % FR
        return [(a + x * m) % FR for x in range(npts)]

    def compute_message(self, rnd, previous_claim):
        phase1 = rnd < self.log_N
        if self.kind in ("sum_axis",) or (self.kind == "exp_sum" and phase1):
            n_co, trimmed = 2, True
        elif (self.kind == "exp_sum" and not phase1) or (self.kind == "recip" and phase1):
            n_co, trimmed = 3, False
        else:
            n_co, trimmed = 4, False
        ev = [0] * n_co
        for i in range(len(self.tabs[0]) // 2):
            cols = [self._ext(t, i, n_co) for t in self.tabs]
            for X in range(n_co):
                t = 1
                for c in cols:
                    t = t * c[X] % FR
                ev[X] = (ev[X] + t) % FR
        assert (ev[0] + ev[1]) % FR == previous_claim % FR
        c = interpolate(ev)
        return from_coeff(c) if trimmed else c

    def ingest_challenge(self, r, rnd):
        self.tabs = [P.bind(t, r, self.order) for t in self.tabs]

    def finals(self):
        n = 2 if self.kind in ("max", "recip") else 1
        return [t[0] for t in self.tabs[:n]]
