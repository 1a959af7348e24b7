"""Generic SumcheckInstanceProver handles (atlas_instance_t) and the one-hot "ra" instances:
RaSumcheckProver (ra_virtual.rs:97-145), BooleanitySumcheckProver (booleanity.rs:169-348),
HammingWeightSumcheckProver (hamming_weight.rs:106-148)."""
import ctypes as C

import numpy as np

from . import U128, _check, _fr, _p, lib

lib.atlas_instance_num_rounds.restype = C.c_size_t
lib.atlas_instance_degree.restype = C.c_size_t


class Instance:
    def __init__(self, handle, keep=()):
        self.h = handle
        self._keep = keep

    def num_rounds(self):
        return lib.atlas_instance_num_rounds(self.h)

    def degree(self):
        return lib.atlas_instance_degree(self.h)

    def compute_message(self, rnd, previous_claim):
        cap = self.degree() + 2
        out = np.zeros((cap, 4), dtype=np.uint64)
        n = C.c_size_t()
        pc = _fr(previous_claim)
        _check(lib.atlas_instance_compute_message(self.h, C.c_size_t(rnd), _p(pc), _p(out), C.c_size_t(cap), C.byref(n)))
        return out[:n.value]

    def ingest_challenge(self, r_u128, rnd):
        u = U128(r_u128 & ((1 << 64) - 1), r_u128 >> 64)
        _check(lib.atlas_instance_ingest_challenge(self.h, C.byref(u), C.c_size_t(rnd)))

    def final_claims(self):
        out = np.zeros((64, 4), dtype=np.uint64)
        n = C.c_size_t()
        _check(lib.atlas_instance_final_claims(self.h, _p(out), C.c_size_t(64), C.byref(n)))
        return out[:n.value]

    def prove(self, input_claim, transcript):
        """Sumcheck::prove. Returns (rows of compressed coefficients, challenges [u128])."""
        n = self.num_rounds(); stride = self.degree() + 1
        comp = np.zeros((max(n, 1), stride, 4), dtype=np.uint64)
        nco = np.zeros(max(n, 1), dtype=np.uint32); ch = np.zeros(2 * max(n, 1), dtype=np.uint64)
        ic = _fr(input_claim)
        _check(lib.atlas_instance_prove(self.h, _p(ic), C.byref(transcript.t), _p(comp), C.c_size_t(stride),
                                        nco.ctypes.data_as(C.c_void_p), _p(ch)))
        return [comp[i, :nco[i]].copy() for i in range(n)], [int(ch[2 * i]) | (int(ch[2 * i + 1]) << 64) for i in range(n)]

    def free(self):
        if self.h:
            lib.atlas_instance_free(self.h)
            self.h = None


def _idx_ptrs(H_indices):
    arrs = [np.ascontiguousarray(a, dtype=np.int32) for a in H_indices]
    return arrs, (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])


def ra_virtual(H_indices, log_k_chunk, r_address_chunks, r_cycle):
    arrs, ptrs = _idx_ptrs(H_indices)
    ch = np.ascontiguousarray(r_address_chunks, dtype=np.uint64); rc = np.ascontiguousarray(r_cycle, dtype=np.uint64)
    h = C.c_void_p()
    _check(lib.atlas_ra_virtual_new(ptrs, C.c_size_t(len(arrs)), C.c_size_t(log_k_chunk), C.c_size_t(len(rc)), _p(ch), _p(rc),
                                    C.byref(h)))
    return Instance(h)


def booleanity(G, H_indices, log_k_chunk, gammas, r_address, r_cycle):
    arrs, ptrs = _idx_ptrs(H_indices)
    G = np.ascontiguousarray(G, dtype=np.uint64); ga = np.ascontiguousarray(gammas, dtype=np.uint64)
    ra = np.ascontiguousarray(r_address, dtype=np.uint64); rc = np.ascontiguousarray(r_cycle, dtype=np.uint64)
    h = C.c_void_p()
    _check(lib.atlas_booleanity_new(_p(G), ptrs, C.c_size_t(len(arrs)), C.c_size_t(log_k_chunk), C.c_size_t(len(rc)), _p(ga),
                                    _p(ra), _p(rc), C.byref(h)))
    return Instance(h)


def hamming_weight(G, log_k_chunk, gamma_powers):
    G = np.ascontiguousarray(G, dtype=np.uint64); gp = np.ascontiguousarray(gamma_powers, dtype=np.uint64)
    h = C.c_void_p()
    _check(lib.atlas_hamming_weight_new(_p(G), C.c_size_t(len(gp)), C.c_size_t(log_k_chunk), _p(gp), C.byref(h)))
    return Instance(h)


def dense_opening(poly, opening_point):
    """DensePolynomialProverOpening over a device polynomial (ownership moves)."""
    pt = np.ascontiguousarray(opening_point, dtype=np.uint64)
    h = C.c_void_p()
    _check(lib.atlas_dense_opening_new(poly.h, _p(pt), C.c_size_t(len(pt)), C.byref(h)))
    poly.h = None
    return Instance(h)


def onehot_opening(nonzero_indices, log_K, r_address, r_cycle):
    idx = np.ascontiguousarray(nonzero_indices, dtype=np.int32)
    ra = np.ascontiguousarray(r_address, dtype=np.uint64); rc = np.ascontiguousarray(r_cycle, dtype=np.uint64)
    h = C.c_void_p()
    _check(lib.atlas_onehot_opening_new(idx.ctypes.data_as(C.c_void_p), C.c_size_t(log_K), C.c_size_t(len(rc)), _p(ra), _p(rc),
                                        C.byref(h)))
    return Instance(h)


def eval_reduction_prove(mle, points, claims, transcript):
    """EvalReductionInstance::prove (evaluation_reduction.rs:91-147). points (N, n, 4).
    Returns (h (len,4), r' (n,4), v' (4,))."""
    pts = np.ascontiguousarray(points, dtype=np.uint64); cl = np.ascontiguousarray(claims, dtype=np.uint64)
    N, n = pts.shape[0], pts.shape[1]
    cap = n * max(N - 1, 1) + 1
    h = np.zeros((cap, 4), dtype=np.uint64); hl = C.c_size_t()
    r = np.zeros((max(n, 1), 4), dtype=np.uint64); c = np.zeros(4, dtype=np.uint64)
    _check(lib.atlas_eval_reduction_prove(mle.h, _p(pts), _p(cl), C.c_size_t(N), C.c_size_t(n), C.byref(transcript.t), _p(h),
                                          C.c_size_t(cap), C.byref(hl), _p(r), _p(c)))
    return h[:hl.value].copy(), r[:n].copy(), c


class DeviceU64:
    """T lookup indices resident in HBM (atlas_u64_upload / atlas_lookup_indices_from_operands); accepted wherever the
    constructors below take `lookup_indices`."""

    def __init__(self, ptr, n):
        self.ptr, self.n = ptr, n

    @classmethod
    def upload(cls, arr):
        arr = np.ascontiguousarray(arr, dtype=np.uint64)
        d = C.c_void_p()
        _check(lib.atlas_u64_upload(arr.ctypes.data_as(C.c_void_p), C.c_size_t(arr.size), C.byref(d)))
        return cls(d, arr.size)

    @classmethod
    def from_operands(cls, left, right=None):
        """compute_lookup_indices_from_operands over device Tensor<i32> operands (TensorI32)."""
        n = int(np.prod(left.shape))
        d = C.c_void_p()
        _check(lib.atlas_lookup_indices_from_operands(left.d, right.d if right is not None else None, C.c_size_t(n), C.byref(d)))
        return cls(d, n)

    def free(self):
        if self.ptr:
            lib.atlas_u64_free(self.ptr)
            self.ptr = None


def _lookup_arg(x):
    """(pointer, keep-alive) for host arrays or DeviceU64"""
    if isinstance(x, DeviceU64):
        return x.ptr, x
    a = np.ascontiguousarray(x, dtype=np.uint64)
    return a.ctypes.data_as(C.c_void_p), a


def ps_shout_relu(lookup_indices, xlen, r_node_output, gamma):
    """ps_read_raf_prover for ReluTable<xlen> (ps_shout/unary.rs:110-148)."""
    idx_p, _keep = _lookup_arg(lookup_indices)
    rn = np.ascontiguousarray(r_node_output, dtype=np.uint64); g = _fr(gamma)
    h = C.c_void_p()
    _check(lib.atlas_ps_shout_relu_new(idx_p, C.c_size_t(len(rn)), C.c_size_t(xlen), _p(rn), _p(g),
                                       C.byref(h)))
    return Instance(h)


def identity_range_check(lookup_indices, log_K, phases, r_node_output):
    """IdentityRCProver::gen (identity_range_check.rs:196-233)."""
    idx_p, _keep = _lookup_arg(lookup_indices)
    rn = np.ascontiguousarray(r_node_output, dtype=np.uint64)
    h = C.c_void_p()
    _check(lib.atlas_identity_range_check_new(idx_p, C.c_size_t(len(rn)), C.c_size_t(log_K),
                                              C.c_size_t(phases), _p(rn), C.byref(h)))
    return Instance(h)


def ps_shout_clamp(lookup_indices, xlen, bound, symmetric, r_node_output, gamma):
    """ps_read_raf_prover for ClampBoundedTable<xlen, bound, symmetric> (lookup_tables/clamp.rs)."""
    idx_p, _keep = _lookup_arg(lookup_indices)
    rn = np.ascontiguousarray(r_node_output, dtype=np.uint64); g = _fr(gamma)
    h = C.c_void_p()
    _check(lib.atlas_ps_shout_clamp_new(idx_p, C.c_size_t(len(rn)), C.c_size_t(xlen), C.c_size_t(bound),
                                        C.c_int(1 if symmetric else 0), _p(rn), _p(g), C.byref(h)))
    return Instance(h)


EW_ADD, EW_SUB, EW_NEG, EW_SQUARE, EW_IFF, EW_MUL, EW_CUBE, EW_DIV, EW_RSQRT, EW_DOT, EW_GATHER, EW_HAMMING_BOOL, EW_TELEPORT_DIV = range(13)


def elementwise(op, operands, r_node_output, constants=None):
    """Element-wise operator prover (ops/add.rs, sub.rs, neg.rs, square.rs, iff.rs, mul.rs, cube.rs) over device
    polynomials (MultilinearPolynomial handles; not consumed)."""
    rn = np.ascontiguousarray(r_node_output, dtype=np.uint64)
    hs = (C.c_void_p * len(operands))(*[o.h for o in operands])
    h = C.c_void_p()
    k = np.ascontiguousarray(constants, dtype=np.uint64).reshape(-1, 4) if constants is not None else None
    _check(lib.atlas_elementwise_new(C.c_int(op), hs, C.c_size_t(len(operands)), _p(rn), C.c_size_t(len(rn)),
                                     _p(k) if k is not None else None, C.c_size_t(0 if k is None else len(k)), C.byref(h)))
    return Instance(h)


SM_EXP_SUM, SM_MAX_INDICATOR, SM_RECIP_MULT, SM_SUM_AXIS = range(4)


def softmax_instance(kind, a, b, log_K, log_N, r):
    """ExpSumProver / MaxIndicatorProver / RecipMultProver (ops/softmax_last_axis/) and SumAxisProver (ops/sum/axis.rs)
    over device polynomials (not consumed)."""
    rr = np.ascontiguousarray(r, dtype=np.uint64) if r is not None else None
    h = C.c_void_p()
    _check(lib.atlas_softmax_instance_new(C.c_int(kind), a.h, b.h if b is not None else None, C.c_size_t(log_K), C.c_size_t(log_N),
                                          _p(rr) if rr is not None else None, C.byref(h)))
    return Instance(h)


def ps_shout_rshift(lookup_indices, xlen, shift, r_node_output, gamma):
    """ps_read_raf_prover for RightShiftTable<xlen> by `shift` bits (lookup_tables/right_shift.rs; Sin/Cos downscale)."""
    idx_p, _keep = _lookup_arg(lookup_indices)
    rn = np.ascontiguousarray(r_node_output, dtype=np.uint64); g = _fr(gamma)
    h = C.c_void_p()
    _check(lib.atlas_ps_shout_rshift_new(idx_p, C.c_size_t(len(rn)), C.c_size_t(xlen), C.c_size_t(shift),
                                         _p(rn), _p(g), C.byref(h)))
    return Instance(h)


def ps_shout_ult(lookup_indices, r_node_output, gamma):
    """binary ps_read_raf_prover with UnsignedLessThanTable<32> (ps_shout/binary.rs:148-200); indices = interleave_bits(x, y)."""
    idx_p, _keep = _lookup_arg(lookup_indices)
    rn = np.ascontiguousarray(r_node_output, dtype=np.uint64); g = _fr(gamma)
    h = C.c_void_p()
    _check(lib.atlas_ps_shout_ult_new(idx_p, C.c_size_t(len(rn)), _p(rn), _p(g), C.byref(h)))
    return Instance(h)


def ra_virtual_from_lookups(lookup_indices, log_K, log_k_chunk, r_address, r_cycle):
    """RaSumcheckProver::gen from the lookup indices (chunks cut on the device, config.rs:73-100)."""
    idx_p, _keep = _lookup_arg(lookup_indices)
    ra = np.ascontiguousarray(r_address, dtype=np.uint64); rc = np.ascontiguousarray(r_cycle, dtype=np.uint64)
    h = C.c_void_p()
    _check(lib.atlas_ra_virtual_from_lookups_new(idx_p, C.c_size_t(len(rc)), C.c_size_t(log_K),
                                                 C.c_size_t(log_k_chunk), _p(ra), _p(rc), C.byref(h)))
    return Instance(h)


def booleanity_from_lookups(G, lookup_indices, log_K, log_k_chunk, gammas, r_address, r_cycle):
    idx_p, _keep = _lookup_arg(lookup_indices)
    G = np.ascontiguousarray(G, dtype=np.uint64); ga = np.ascontiguousarray(gammas, dtype=np.uint64)
    ra = np.ascontiguousarray(r_address, dtype=np.uint64); rc = np.ascontiguousarray(r_cycle, dtype=np.uint64)
    h = C.c_void_p()
    _check(lib.atlas_booleanity_from_lookups_new(_p(G), idx_p, C.c_size_t(len(rc)), C.c_size_t(log_K),
                                                 C.c_size_t(log_k_chunk), _p(ga), _p(ra), _p(rc), C.byref(h)))
    return Instance(h)
