"""ctypes binding of oracle/ra.c (RaVirtual / Booleanity / HammingWeight instances).
TEST INFRASTRUCTURE ONLY, like orc.py."""
import ctypes as C

import numpy as np

from . import orc

RA_VIRTUAL, BOOLEANITY, HAMMING = 2, 3, 4
_STATE_BYTES = 16384         # >= sizeof of any orc_* state struct


def _idx_ptrs(H_indices):
    arrs = [np.ascontiguousarray(a, dtype=np.int32) for a in H_indices]
    ptrs = (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
    return arrs, ptrs


class Instance:
    def __init__(self, kind, n_rounds):
        self.kind, self.n_rounds = kind, n_rounds
        self.st = C.create_string_buffer(_STATE_BYTES)
        self.keep = []

    def message(self, rnd, claim):
        out = orc.fr_array(40)
        orc.lib.orc_ra_inst_message.restype = C.c_size_t
        n = orc.lib.orc_ra_inst_message(C.c_int(self.kind), self.st, C.c_size_t(rnd), orc._p(np.ascontiguousarray(claim)), orc._p(out))
        return out[:n]

    def ingest(self, rnd, r_fr):
        orc.lib.orc_ra_inst_ingest(C.c_int(self.kind), self.st, C.c_size_t(rnd), orc._p(np.ascontiguousarray(r_fr)))

    def prove(self, claim, t, stride=20):
        n = self.n_rounds
        comp = orc.fr_array(max(n, 1) * stride); nco = np.zeros(max(n, 1), dtype=np.uint32)
        ch = np.zeros(2 * max(n, 1), dtype=np.uint64)
        claim = np.ascontiguousarray(claim, dtype=np.uint64).reshape(1, 4)
        orc.lib.orc_ra_inst_prove(C.c_int(self.kind), self.st, C.c_size_t(n), orc._p(claim), C.byref(t), orc._p(comp),
                                  C.c_size_t(stride), nco.ctypes.data_as(C.c_void_p), orc._p(ch))
        comp = comp.reshape(max(n, 1), stride, 4)
        return [comp[i, :nco[i]].copy() for i in range(n)], orc._u128_list(ch, n)


def ra_virtual(H_indices, log_k, r_address_chunks, r_cycle):
    """r_address_chunks: (d, log_k, 4) Fr; r_cycle: (log_T, 4) Fr."""
    d = len(H_indices); log_T = len(r_cycle)
    I = Instance(RA_VIRTUAL, log_T)
    arrs, ptrs = _idx_ptrs(H_indices)
    ch = np.ascontiguousarray(r_address_chunks, dtype=np.uint64); rc = np.ascontiguousarray(r_cycle, dtype=np.uint64)
    I.keep = [arrs, ptrs, ch, rc]
    orc.lib.orc_ra_virtual_init(I.st, ptrs, C.c_size_t(d), C.c_size_t(log_k), C.c_size_t(log_T), orc._p(ch), orc._p(rc))
    return I


def booleanity(G, H_indices, log_k, gammas, r_address, r_cycle):
    d = len(H_indices); log_T = len(r_cycle)
    I = Instance(BOOLEANITY, log_k + log_T)
    arrs, ptrs = _idx_ptrs(H_indices)
    G = np.ascontiguousarray(G, dtype=np.uint64); ga = np.ascontiguousarray(gammas, dtype=np.uint64)
    ra = np.ascontiguousarray(r_address, dtype=np.uint64); rc = np.ascontiguousarray(r_cycle, dtype=np.uint64)
    I.keep = [arrs, ptrs, G, ga, ra, rc]
    orc.lib.orc_booleanity_init(I.st, orc._p(G), ptrs, C.c_size_t(d), C.c_size_t(log_k), C.c_size_t(log_T), orc._p(ga),
                                orc._p(ra), orc._p(rc))
    return I


def hamming(G, log_k, gamma_powers):
    G = np.ascontiguousarray(G, dtype=np.uint64); gp = np.ascontiguousarray(gamma_powers, dtype=np.uint64)
    d = len(gp)
    I = Instance(HAMMING, log_k)
    I.keep = [G, gp]
    orc.lib.orc_hamming_init(I.st, orc._p(G), C.c_size_t(d), C.c_size_t(log_k), orc._p(gp))
    return I


def ra_G(H_indices, log_k, r_cycle):
    """compute_ra_evals-style G_i[k] = sum_{j: idx_i[j] = k} eq(r_cycle, j) (shout.rs:550-598)."""
    E = orc.eq_evals(np.ascontiguousarray(r_cycle))
    K = 1 << log_k
    out = orc.fr_array(len(H_indices) * K).reshape(len(H_indices), K, 4)
    for i, idx in enumerate(H_indices):
        for j, k in enumerate(idx):
            if k >= 0:
                out[i, k] = orc.fr_add_arr(out[i, k], E[j])
    return out


DENSE_OPENING, ONEHOT_OPENING = 5, 6


def dense_opening(poly, point):
    """DensePolynomialProverOpening: poly (2^n, 4) Fr, point (n, 4) Fr."""
    poly = np.ascontiguousarray(poly, dtype=np.uint64); point = np.ascontiguousarray(point, dtype=np.uint64)
    I = Instance(DENSE_OPENING, len(point))
    I.keep = [poly, point]
    orc.lib.orc_dense_opening_init(I.st, orc._p(poly), C.c_size_t(len(point)), orc._p(point))
    return I


def onehot_opening(idx, log_K, r_address, r_cycle):
    idx = np.ascontiguousarray(idx, dtype=np.int32)
    ra = np.ascontiguousarray(r_address, dtype=np.uint64); rc = np.ascontiguousarray(r_cycle, dtype=np.uint64)
    I = Instance(ONEHOT_OPENING, log_K + len(rc))
    I.keep = [idx, ra, rc]
    orc.lib.orc_onehot_opening_init(I.st, idx.ctypes.data_as(C.c_void_p), C.c_size_t(log_K), C.c_size_t(len(rc)), orc._p(ra), orc._p(rc))
    return I


def eval_reduction_prove(mle, points, claims, t):
    """EvalReductionInstance::prove. points (N, n, 4). Returns (h (len,4), r (n,4), claim (4,))."""
    mle = np.ascontiguousarray(mle, dtype=np.uint64); pts = np.ascontiguousarray(points, dtype=np.uint64)
    cl = np.ascontiguousarray(claims, dtype=np.uint64)
    N, n = pts.shape[0], pts.shape[1]
    h = orc.fr_array(n * max(N - 1, 1) + 1); hl = C.c_size_t(); r = orc.fr_array(max(n, 1)); c = orc.fr_array(1)
    orc.lib.orc_eval_reduction_prove.restype = C.c_int
    rc = orc.lib.orc_eval_reduction_prove(orc._p(mle), C.c_size_t(n), orc._p(pts), orc._p(cl), C.c_size_t(N), C.byref(t),
                                          orc._p(h), C.byref(hl), orc._p(r), orc._p(c))
    assert rc == 0
    return h[:hl.value].copy(), r[:n].copy(), c[0].copy()


PS_RELU = 7


def ps_relu(lookup_indices, N, r_node, gamma):
    """Unary prefix-suffix read-raf prover with the ReLU table (ps_shout/unary.rs:110-148)."""
    idx = np.ascontiguousarray(lookup_indices, dtype=np.uint64)
    rn = np.ascontiguousarray(r_node, dtype=np.uint64); g = np.ascontiguousarray(gamma, dtype=np.uint64).reshape(1, 4)
    I = Instance(PS_RELU, N + len(rn))
    I.keep = [idx, rn, g]
    orc.lib.orc_ps_relu_init(I.st, idx.ctypes.data_as(C.c_void_p), C.c_size_t(N), C.c_size_t(len(rn)), orc._p(rn), orc._p(g))
    return I


def ps_rshift(lookup_indices, N, shift, r_node, gamma):
    """The same unary read-raf prover over RightShiftTable<N> (lookup_tables/right_shift.rs), D = shift."""
    idx = np.ascontiguousarray(lookup_indices, dtype=np.uint64)
    rn = np.ascontiguousarray(r_node, dtype=np.uint64); g = np.ascontiguousarray(gamma, dtype=np.uint64).reshape(1, 4)
    I = Instance(PS_RELU, N + len(rn))
    I.keep = [idx, rn, g]
    orc.lib.orc_ps_rshift_init(I.st, idx.ctypes.data_as(C.c_void_p), C.c_size_t(N), C.c_size_t(shift), C.c_size_t(len(rn)), orc._p(rn), orc._p(g))
    return I


ELEMENTWISE = 11
EW_ADD, EW_SUB, EW_NEG, EW_SQUARE, EW_IFF, EW_MUL, EW_CUBE, EW_DIV, EW_RSQRT, EW_DOT, EW_GATHER, EW_HAMMING_BOOL, EW_TELEPORT_DIV = range(13)


def elementwise(op, operands, r_node_output, constants=None):
    """Element-wise operator prover (ops/add.rs, sub.rs, neg.rs, square.rs, mul.rs, iff.rs, cube.rs)."""
    ops = [np.ascontiguousarray(o, dtype=np.uint64) for o in operands]
    rn = np.ascontiguousarray(r_node_output, dtype=np.uint64)
    ptrs = (C.c_void_p * len(ops))(*[o.ctypes.data for o in ops])
    I = Instance(ELEMENTWISE, len(rn))
    k = np.ascontiguousarray(constants, dtype=np.uint64).reshape(-1, 4) if constants is not None else np.zeros((0, 4), dtype=np.uint64)
    I.keep = [ops, ptrs, rn, k]
    orc.lib.orc_elementwise_init(I.st, C.c_int(op), ptrs, C.c_size_t(len(ops)), C.c_size_t(len(rn)), orc._p(rn),
                                 orc._p(k) if len(k) else None, C.c_size_t(len(k)))
    I.n_ops = len(ops)

    def finals():
        out = orc.fr_array(I.n_ops)
        orc.lib.orc_elementwise_finals(I.st, orc._p(out))
        return out
    I.finals = finals
    return I


SOFTMAX = 12
SM_EXP_SUM, SM_MAX_INDICATOR, SM_RECIP_MULT, SM_SUM_AXIS = range(4)


def softmax(kind, a, b, log_K, log_N, r):
    """ExpSumProver / MaxIndicatorProver / RecipMultProver (ops/softmax_last_axis/) and SumAxisProver (ops/sum/axis.rs)."""
    a = np.ascontiguousarray(a, dtype=np.uint64)
    b = np.ascontiguousarray(b, dtype=np.uint64) if b is not None else None
    r = np.ascontiguousarray(r, dtype=np.uint64) if r is not None else np.zeros((0, 4), dtype=np.uint64)
    I = Instance(SOFTMAX, log_K + log_N)
    I.keep = [a, b, r]
    orc.lib.orc_softmax_init(I.st, C.c_int(kind), orc._p(a), orc._p(b) if b is not None else None, C.c_size_t(log_K), C.c_size_t(log_N),
                             orc._p(r) if len(r) else None)
    n_fin = 2 if b is not None else 1

    def finals():
        out = orc.fr_array(2)
        orc.lib.orc_softmax_finals(I.st, orc._p(out))
        return out[:n_fin]
    I.finals = finals
    return I


PS_IDENTITY = 8


def ps_identity(lookup_indices, log_K, phases, r_node):
    """IdentityRCProver (identity_range_check.rs:196-420)."""
    idx = np.ascontiguousarray(lookup_indices, dtype=np.uint64); rn = np.ascontiguousarray(r_node, dtype=np.uint64)
    I = Instance(PS_IDENTITY, log_K + len(rn))
    I.keep = [idx, rn]
    orc.lib.orc_ps_identity_init(I.st, idx.ctypes.data_as(C.c_void_p), C.c_size_t(log_K), C.c_size_t(phases), C.c_size_t(len(rn)), orc._p(rn))
    return I


PS_CLAMP = 9


def ps_clamp(lookup_indices, N, bound, symmetric, r_node, gamma):
    """Unary read-raf prover with ClampBoundedTable<N, bound, symmetric> (lookup_tables/clamp.rs)."""
    idx = np.ascontiguousarray(lookup_indices, dtype=np.uint64)
    rn = np.ascontiguousarray(r_node, dtype=np.uint64); g = np.ascontiguousarray(gamma, dtype=np.uint64).reshape(1, 4)
    I = Instance(PS_CLAMP, N + len(rn))
    I.keep = [idx, rn, g]
    orc.lib.orc_ps_clamp_init(I.st, idx.ctypes.data_as(C.c_void_p), C.c_size_t(N), C.c_size_t(bound), C.c_int(1 if symmetric else 0),
                              C.c_size_t(len(rn)), orc._p(rn), orc._p(g))
    return I


PS_ULT = 10


def ps_ult(lookup_indices, r_node, gamma):
    """Binary read-raf prover with UnsignedLessThanTable<32> (ps_shout/binary.rs:148-200)."""
    idx = np.ascontiguousarray(lookup_indices, dtype=np.uint64)
    rn = np.ascontiguousarray(r_node, dtype=np.uint64); g = np.ascontiguousarray(gamma, dtype=np.uint64).reshape(1, 4)
    I = Instance(PS_ULT, 64 + len(rn))
    I.keep = [idx, rn, g]
    orc.lib.orc_ps_ult_init(I.st, idx.ctypes.data_as(C.c_void_p), C.c_size_t(len(rn)), orc._p(rn), orc._p(g))
    return I


def interleave(x, y):
    """interleave_bits (utils/mod.rs:146-164): x on the odd bit positions, y on the even ones."""
    out = 0
    for i in range(32):
        out |= ((int(x) >> i) & 1) << (2 * i + 1)
        out |= ((int(y) >> i) & 1) << (2 * i)
    return out


SHOUT = 13


def shout_read_raf(lookup_indices, table, log_K, r_cycle, gamma):
    """ReadRafProver (joltworks shout.rs:193-277) as a batch member: G[k] = sum_{j : idx_j = k} eq(r_cycle, j), then
    sum_k G[k] (table[k] + gamma k), HighToLow, degree 2.  final() = G(r)."""
    idx = np.ascontiguousarray(lookup_indices, dtype=np.uint64)
    tab = np.ascontiguousarray(table, dtype=np.int32)
    assert len(tab) == 1 << log_K
    E = orc.eq_evals(np.ascontiguousarray(r_cycle, dtype=np.uint64))
    G = orc.fr_array(1 << log_K)
    orc.lib.orc_shout_G(idx.ctypes.data_as(C.c_void_p), C.c_size_t(len(idx)), C.c_size_t(log_K), orc._p(E), orc._p(G))
    g = np.ascontiguousarray(gamma, dtype=np.uint64).reshape(1, 4)
    I = Instance(SHOUT, log_K)
    I.keep = [idx, tab, G, g]
    orc.lib.orc_shout_inst_init(I.st, orc._p(G), tab.ctypes.data_as(C.c_void_p), C.c_size_t(log_K), orc._p(g))

    def final():
        out = orc.fr_array(1)
        orc.lib.orc_shout_inst_final(I.st, orc._p(out))
        return out[0]
    I.final = final
    return I


def ra_claim(lookup_indices, log_K, point):
    """ra(r_address | r_cycle) of the one-hot read-address polynomial of the lookups (point = log_K address coordinates, then log_T cycle ones)"""
    idx = np.ascontiguousarray(lookup_indices, dtype=np.uint64)
    pt = np.ascontiguousarray(point, dtype=np.uint64)
    log_T = len(pt) - log_K
    assert len(idx) == 1 << log_T
    out = orc.fr_array(1)
    orc.lib.orc_ra_claim(idx.ctypes.data_as(C.c_void_p), C.c_size_t(log_T), C.c_size_t(log_K), orc._p(pt[:log_K].copy()), orc._p(pt[log_K:].copy()), orc._p(out))
    return out[0]
