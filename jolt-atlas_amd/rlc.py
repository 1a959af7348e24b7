"""build_materialized_rlc (joltworks/src/poly/rlc_polynomial.rs:13-78) over device polynomials."""
import ctypes as C

import numpy as np

from . import MultilinearPolynomial, _check, _fr, lib


class _Dense(C.Structure):
    _fields_ = [("poly", C.c_void_p), ("coeff", C.c_uint64 * 4)]


class _OneHot(C.Structure):
    _fields_ = [("k", C.c_void_p), ("T", C.c_size_t), ("K", C.c_size_t), ("coeff", C.c_uint64 * 4), ("k_on_device", C.c_int)]


def build_materialized_rlc(dense, onehot):
    """dense: list of (MultilinearPolynomial, coeff Fr); onehot: list of (k int32 array with
    negative = None, K, coeff Fr).  Returns the joint LargeScalars polynomial (device)."""
    nd, no = len(dense), len(onehot)
    D = (_Dense * max(nd, 1))()
    for j, (p, c) in enumerate(dense):
        D[j].poly = p.h
        c = _fr(c).reshape(4)
        for q in range(4):
            D[j].coeff[q] = int(c[q])
    O = (_OneHot * max(no, 1))()
    keep = []
    for j, (k, K, c) in enumerate(onehot):
        k = np.ascontiguousarray(k, dtype=np.int32); keep.append(k)
        O[j].k = k.ctypes.data; O[j].T = len(k); O[j].K = K; O[j].k_on_device = 0
        c = _fr(c).reshape(4)
        for q in range(4):
            O[j].coeff[q] = int(c[q])
    h = C.c_void_p()
    _check(lib.atlas_rlc_build(D, C.c_size_t(nd), O, C.c_size_t(no), C.byref(h)))
    return MultilinearPolynomial(h)
