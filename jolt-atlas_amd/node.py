"""Operator-node provers composed in the library (atlas_prove_einsum_node)."""
import ctypes as C

import numpy as np

from . import _check, _fr, _p, lib


class _Node(C.Structure):
    _fields_ = [("m", C.c_size_t), ("k", C.c_size_t), ("n", C.c_size_t), ("scale_bits", C.c_uint32)]


def prove_einsum_node(tA, tB, m, k, n, scale_bits, r_node_output, transcript, output_claim=None):
    """Einsum::prove for a fused-rescale mk,kn->mn node.  tA, tB: TensorI32.  Returns (proof_bytes list of 5, claims (c,4), stage_ms (6,))."""
    nd = _Node(m, k, n, scale_bits)
    rn = np.ascontiguousarray(r_node_output, dtype=np.uint64)
    cap = 1 << 20
    buf = (C.c_uint8 * cap)(); ln = C.c_size_t(); lens = (C.c_size_t * 5)()
    claims = np.zeros((256, 4), dtype=np.uint64); nc = C.c_size_t(); st = (C.c_double * 6)()
    oc = _p(_fr(output_claim)) if output_claim is not None else None
    _check(lib.atlas_prove_einsum_node(C.byref(nd), tA.d, tB.d, _p(rn), oc, C.byref(transcript.t), buf, C.c_size_t(cap), C.byref(ln), lens,
                                       _p(claims), C.c_size_t(256), C.byref(nc), None, st))
    raw = bytes(buf[:ln.value])
    out, o = [], 0
    for i in range(5):
        out.append(raw[o:o + lens[i]]); o += lens[i]
    return out, claims[:nc.value].copy(), np.array(list(st))


def prove_relu_node(tX, log_T, r_node_output, transcript, output_claim=None):
    """ReLU::prove for one node.  tX: TensorI32 of 2^log_T values.  Returns (proof_bytes list of 2, claims (c,4), stage_ms (3,))."""
    rn = np.ascontiguousarray(r_node_output, dtype=np.uint64)
    cap = 1 << 20
    buf = (C.c_uint8 * cap)(); ln = C.c_size_t(); lens = (C.c_size_t * 2)()
    claims = np.zeros((256, 4), dtype=np.uint64); nc = C.c_size_t(); st = (C.c_double * 3)()
    oc = _p(_fr(output_claim)) if output_claim is not None else None
    _check(lib.atlas_prove_relu_node(tX.d, C.c_size_t(log_T), _p(rn), oc, C.byref(transcript.t), buf, C.c_size_t(cap), C.byref(ln), lens,
                                     _p(claims), C.c_size_t(256), C.byref(nc), None, st))
    raw = bytes(buf[:ln.value])
    return [raw[:lens[0]], raw[lens[0]:lens[0] + lens[1]]], claims[:nc.value].copy(), np.array(list(st))


def prove_mul_node(tL, tR, log_T, scale_bits, r_node_output, transcript, output_claim=None):
    """Mul::prove (fused rescale) for one node.  tL, tR: TensorI32 of 2^log_T values.  Returns (proof_bytes list of 5, claims (c,4), stage_ms (6,))."""
    rn = np.ascontiguousarray(r_node_output, dtype=np.uint64)
    cap = 1 << 20
    buf = (C.c_uint8 * cap)(); ln = C.c_size_t(); lens = (C.c_size_t * 5)()
    claims = np.zeros((256, 4), dtype=np.uint64); nc = C.c_size_t(); st = (C.c_double * 6)()
    oc = _p(_fr(output_claim)) if output_claim is not None else None
    _check(lib.atlas_prove_mul_node(tL.d, tR.d, C.c_size_t(log_T), C.c_uint32(scale_bits), _p(rn), oc, C.byref(transcript.t), buf, C.c_size_t(cap),
                                    C.byref(ln), lens, _p(claims), C.c_size_t(256), C.byref(nc), None, st))
    raw = bytes(buf[:ln.value])
    out, o = [], 0
    for i in range(5):
        out.append(raw[o:o + lens[i]]); o += lens[i]
    return out, claims[:nc.value].copy(), np.array(list(st))


def _node_call(fn, args_head, n_proofs, n_stages, r_node_output, transcript, output_claim):
    rn = np.ascontiguousarray(r_node_output, dtype=np.uint64)
    cap = 1 << 20
    buf = (C.c_uint8 * cap)(); ln = C.c_size_t(); lens = (C.c_size_t * n_proofs)()
    claims = np.zeros((256, 4), dtype=np.uint64); nc = C.c_size_t(); st = (C.c_double * n_stages)()
    oc = _p(_fr(output_claim)) if output_claim is not None else None
    _check(fn(*args_head, _p(rn), oc, C.byref(transcript.t), buf, C.c_size_t(cap), C.byref(ln), lens, _p(claims), C.c_size_t(256), C.byref(nc), None, st))
    raw = bytes(buf[:ln.value])
    out, o = [], 0
    for i in range(n_proofs):
        out.append(raw[o:o + lens[i]]); o += lens[i]
    return out, claims[:nc.value].copy(), np.array(list(st))


def prove_square_node(tX, log_T, scale_bits, r_node_output, transcript, output_claim=None):
    """Square::prove (fused rescale) for one node: 5 proofs."""
    return _node_call(lib.atlas_prove_square_node, (tX.d, C.c_size_t(log_T), C.c_uint32(scale_bits)), 5, 6, r_node_output, transcript, output_claim)


def prove_addsub_node(tL, tR, log_T, subtract, r_node_output, transcript, output_claim=None):
    """Add::prove / Sub::prove for one node: 2 proofs (clamp lookup, one-hot checks) + the operand tie."""
    return _node_call(lib.atlas_prove_addsub_node, (tL.d, tR.d, C.c_size_t(log_T), C.c_int(1 if subtract else 0)), 2, 3, r_node_output, transcript, output_claim)


def prove_standard_node(op, tensors, log_T, r_node_output, output_claim, transcript):
    """And (op = 5, ATLAS_EW_MUL) / Iff (op = 4) node: one sumcheck + the operand claims.  Returns (proof bytes, claims (c,4))."""
    rn = np.ascontiguousarray(r_node_output, dtype=np.uint64)
    ptrs = (C.c_void_p * len(tensors))(*[C.cast(t_.d, C.c_void_p) for t_ in tensors])
    cap = 1 << 18
    buf = (C.c_uint8 * cap)(); ln = C.c_size_t(); lens = (C.c_size_t * 1)()
    claims = np.zeros((8, 4), dtype=np.uint64); nc = C.c_size_t()
    _check(lib.atlas_prove_standard_node(C.c_int(op), ptrs, C.c_size_t(len(tensors)), C.c_size_t(log_T), _p(rn), _p(_fr(output_claim)), C.byref(transcript.t),
                                         buf, C.c_size_t(cap), C.byref(ln), lens, _p(claims), C.c_size_t(8), C.byref(nc)))
    return bytes(buf[:ln.value]), claims[:nc.value].copy()
