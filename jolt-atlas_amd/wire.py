"""ark-serialize images of the proof pieces and the SRS file (atlas_*_serialize / atlas_srs_load_file)."""
import ctypes as C

import numpy as np

from . import G1_DTYPE, SRS, _check, _fr, _p, lib


def fr_to_bytes(fr):
    out = (C.c_uint8 * 32)()
    _check(lib.atlas_fr_to_bytes(_p(_fr(fr)), out))
    return bytes(out)


def fr_from_bytes(b):
    out = np.zeros(4, dtype=np.uint64)
    _check(lib.atlas_fr_from_bytes((C.c_uint8 * 32)(*b), _p(out)))
    return out


def g1_to_bytes(pt, compressed=True):
    a = np.ascontiguousarray(np.asarray(pt, dtype=G1_DTYPE).reshape(1))
    out = (C.c_uint8 * (32 if compressed else 64))()
    f = lib.atlas_g1_to_bytes_compressed if compressed else lib.atlas_g1_to_bytes_uncompressed
    _check(f(a.ctypes.data_as(C.c_void_p), out))
    return bytes(out)


def g1_from_bytes(b):
    out = np.zeros(1, dtype=G1_DTYPE)
    _check(lib.atlas_g1_from_bytes_compressed((C.c_uint8 * 32)(*b), out.ctypes.data_as(C.c_void_p)))
    return out[0]


def sumcheck_proof_to_bytes(rows):
    """rows: list of (k_i, 4) arrays (compressed coefficients per round)."""
    n = len(rows)
    stride = max([len(r) for r in rows] + [1])
    flat = np.zeros((max(n, 1), stride, 4), dtype=np.uint64)
    nco = np.zeros(max(n, 1), dtype=np.uint32)
    for i, r in enumerate(rows):
        flat[i, :len(r)] = r; nco[i] = len(r)
    ln = C.c_size_t()
    _check(lib.atlas_sumcheck_proof_serialize(_p(flat), C.c_size_t(stride), nco.ctypes.data_as(C.c_void_p), C.c_size_t(n), None, C.c_size_t(0), C.byref(ln)))
    out = (C.c_uint8 * ln.value)()
    _check(lib.atlas_sumcheck_proof_serialize(_p(flat), C.c_size_t(stride), nco.ctypes.data_as(C.c_void_p), C.c_size_t(n), out, C.c_size_t(ln.value), C.byref(ln)))
    return bytes(out)


def sumcheck_proof_from_bytes(b, max_rounds=64, stride=40):
    rows = np.zeros((max_rounds, stride, 4), dtype=np.uint64); nco = np.zeros(max_rounds, dtype=np.uint32)
    n = C.c_size_t(); used = C.c_size_t()
    buf = (C.c_uint8 * len(b))(*b)
    _check(lib.atlas_sumcheck_proof_deserialize(buf, C.c_size_t(len(b)), _p(rows), C.c_size_t(stride), nco.ctypes.data_as(C.c_void_p),
                                                C.c_size_t(max_rounds), C.byref(n), C.byref(used)))
    return [rows[i, :nco[i]].copy() for i in range(n.value)], used.value


def hyperkzg_proof_to_bytes(com, w, v):
    """com: (ell-1,) G1, w: (3,) G1, v: (3, ell, 4) Fr."""
    com = np.ascontiguousarray(com, dtype=G1_DTYPE); w = np.ascontiguousarray(w, dtype=G1_DTYPE)
    v = np.ascontiguousarray(v, dtype=np.uint64).reshape(3, -1, 4)
    ell = v.shape[1]
    ln = C.c_size_t()
    args = (com.ctypes.data_as(C.c_void_p) if len(com) else None, C.c_size_t(len(com)), w.ctypes.data_as(C.c_void_p), _p(v), C.c_size_t(ell))
    _check(lib.atlas_hyperkzg_proof_serialize(*args, None, C.c_size_t(0), C.byref(ln)))
    out = (C.c_uint8 * ln.value)()
    _check(lib.atlas_hyperkzg_proof_serialize(*args, out, C.c_size_t(ln.value), C.byref(ln)))
    return bytes(out)


def hyperkzg_proof_from_bytes(b, cap_ell=40):
    com = np.zeros(cap_ell, dtype=G1_DTYPE); w = np.zeros(3, dtype=G1_DTYPE); v = np.zeros((3 * cap_ell, 4), dtype=np.uint64)
    nc = C.c_size_t(); ell = C.c_size_t(); used = C.c_size_t()
    buf = (C.c_uint8 * len(b))(*b)
    _check(lib.atlas_hyperkzg_proof_deserialize(buf, C.c_size_t(len(b)), com.ctypes.data_as(C.c_void_p), C.c_size_t(cap_ell), C.byref(nc),
                                                w.ctypes.data_as(C.c_void_p), _p(v), C.c_size_t(cap_ell), C.byref(ell), C.byref(used)))
    return com[:nc.value].copy(), w, v[:3 * ell.value].reshape(3, ell.value, 4).copy()


def srs_load_file(path, max_points=0):
    h = C.c_void_p()
    _check(lib.atlas_srs_load_file(str(path).encode(), C.c_size_t(max_points), C.byref(h)))
    return SRS(h)


def srs_save_file(srs, path):
    _check(lib.atlas_srs_save_file(srs.h, str(path).encode()))
