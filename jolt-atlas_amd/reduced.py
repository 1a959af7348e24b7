"""ONNXProof::prove_reduced_openings (jolt-atlas-core/src/onnx_proof/prover.rs:141-176)."""
import ctypes as C

import numpy as np

from . import G1_DTYPE, U128, _check, _fr, _p, lib


class _Opening(C.Structure):
    _fields_ = [("kind", C.c_int), ("poly", C.c_void_p), ("n", C.c_size_t), ("k", C.c_void_p), ("log_K", C.c_size_t),
                ("log_T", C.c_size_t), ("point", C.c_void_p), ("claim", C.c_uint64 * 4), ("d_lookups", C.c_void_p), ("chunk_shift", C.c_size_t)]


def prove_reduced_openings(openings, srs, transcript):
    """openings: list of dicts {"poly": MultilinearPolynomial, "point": (n,4), "claim": (4,)} (dense) or
    {"k": int32 array | "d_lookups": device address + "chunk_shift", "log_K": int, "r_address": (log_K,4), "r_cycle": (log_T,4), "claim": (4,)} (one-hot),
    in CommittedPoly order.  Returns (rows, challenges, sumcheck_claims, com, w, v)."""
    n = len(openings)
    arr = (_Opening * n)()
    keep = []
    max_rounds = 0
    for i, o in enumerate(openings):
        c = _fr(o["claim"]).reshape(4)
        for q in range(4):
            arr[i].claim[q] = int(c[q])
        if "poly" in o:
            pt = np.ascontiguousarray(o["point"], dtype=np.uint64); keep.append(pt)
            arr[i].kind = 0; arr[i].poly = o["poly"].h; arr[i].n = len(pt); arr[i].point = pt.ctypes.data
            max_rounds = max(max_rounds, len(pt))
        else:
            pt = np.ascontiguousarray(np.concatenate([o["r_address"], o["r_cycle"]]), dtype=np.uint64); keep.append(pt)
            arr[i].kind = 1; arr[i].log_K = o["log_K"]; arr[i].log_T = len(o["r_cycle"])
            if "d_lookups" in o:                       # device-resident lookup indices (int address) + the chunk's shift
                arr[i].d_lookups = o["d_lookups"]; arr[i].chunk_shift = o["chunk_shift"]
            else:
                k = np.ascontiguousarray(o["k"], dtype=np.int32); keep.append(k)
                arr[i].k = k.ctypes.data
            arr[i].point = pt.ctypes.data
            max_rounds = max(max_rounds, o["log_K"] + len(o["r_cycle"]))
    rows = np.zeros((max_rounds, 3, 4), dtype=np.uint64); nco = np.zeros(max_rounds, dtype=np.uint32)
    ch = np.zeros(2 * max_rounds, dtype=np.uint64); mr = C.c_size_t()
    claims = np.zeros((n, 4), dtype=np.uint64)
    com = np.zeros(max(max_rounds - 1, 1), dtype=G1_DTYPE); w = np.zeros(3, dtype=G1_DTYPE)
    v = np.zeros((3 * max_rounds, 4), dtype=np.uint64)
    _check(lib.atlas_prove_reduced_openings(arr, C.c_size_t(n), srs.h, C.byref(transcript.t), _p(rows), nco.ctypes.data_as(C.c_void_p),
                                            _p(ch), C.byref(mr), _p(claims), com.ctypes.data_as(C.c_void_p),
                                            w.ctypes.data_as(C.c_void_p), _p(v)))
    assert mr.value == max_rounds
    return ([rows[i, :nco[i]].copy() for i in range(max_rounds)], [int(ch[2 * i]) | (int(ch[2 * i + 1]) << 64) for i in range(max_rounds)],
            claims, com[:max_rounds - 1], w, v.reshape(3, max_rounds, 4))


class ProverOpeningAccumulator:
    """ProverOpeningAccumulator (opening_proof.rs:195-643) over atlas_accumulator_*; keys are u64 in the Ord of the Rust keys."""

    def __init__(self):
        self.h = C.c_void_p()
        _check(lib.atlas_accumulator_new(C.byref(self.h)))
        self._keep = []

    def append_dense(self, transcript, opening_key, poly_key, poly, point, claim):
        pt = np.ascontiguousarray(point, dtype=np.uint64); c = _fr(claim)
        self._keep.append(poly)
        _check(lib.atlas_accumulator_append_dense(self.h, C.byref(transcript.t), C.c_uint64(opening_key), C.c_uint64(poly_key), poly.h,
                                                  _p(pt), C.c_size_t(len(pt)), _p(c)))

    def append_sparse(self, transcript, opening_keys, poly_keys, index_rows, log_K, r_address, r_cycle, claims):
        rows = [np.ascontiguousarray(r, dtype=np.int32) for r in index_rows]
        R = len(rows)
        ptrs = (C.c_void_p * R)(*[r.ctypes.data for r in rows])
        ok = (C.c_uint64 * R)(*opening_keys); pk = (C.c_uint64 * R)(*poly_keys)
        ra = np.ascontiguousarray(r_address, dtype=np.uint64); rc = np.ascontiguousarray(r_cycle, dtype=np.uint64)
        cl = np.ascontiguousarray(np.stack([_fr(c).reshape(4) for c in claims]), dtype=np.uint64)
        _check(lib.atlas_accumulator_append_sparse(self.h, C.byref(transcript.t), ok, pk, ptrs, C.c_size_t(R), C.c_size_t(log_K),
                                                   C.c_size_t(len(rc)), _p(ra), _p(rc), _p(cl)))

    def append_virtual(self, transcript, opening_key, point, claim):
        pt = np.ascontiguousarray(point, dtype=np.uint64).reshape(-1, 4); c = _fr(claim)
        _check(lib.atlas_accumulator_append_virtual(self.h, C.byref(transcript.t), C.c_uint64(opening_key), _p(pt), C.c_size_t(len(pt)), _p(c)))

    def get(self, opening_key):
        pt = np.zeros((64, 4), dtype=np.uint64); n = C.c_size_t(); c = np.zeros(4, dtype=np.uint64)
        _check(lib.atlas_accumulator_get(self.h, C.c_uint64(opening_key), _p(pt), C.c_size_t(64), C.byref(n), _p(c)))
        return pt[:n.value].copy(), c

    def prove_reduced_openings(self, srs, transcript):
        lib.atlas_accumulator_max_rounds.restype = C.c_size_t
        lib.atlas_accumulator_num_committed.restype = C.c_size_t
        max_rounds = lib.atlas_accumulator_max_rounds(self.h); n = lib.atlas_accumulator_num_committed(self.h)
        rows = np.zeros((max_rounds, 3, 4), dtype=np.uint64); nco = np.zeros(max_rounds, dtype=np.uint32)
        ch = np.zeros(2 * max_rounds, dtype=np.uint64); mr = C.c_size_t()
        claims = np.zeros((n, 4), dtype=np.uint64)
        com = np.zeros(max(max_rounds - 1, 1), dtype=G1_DTYPE); w = np.zeros(3, dtype=G1_DTYPE)
        v = np.zeros((3 * max_rounds, 4), dtype=np.uint64)
        _check(lib.atlas_accumulator_prove_reduced_openings(self.h, srs.h, C.byref(transcript.t), _p(rows), nco.ctypes.data_as(C.c_void_p),
                                                            _p(ch), C.byref(mr), _p(claims), com.ctypes.data_as(C.c_void_p),
                                                            w.ctypes.data_as(C.c_void_p), _p(v)))
        return ([rows[i, :nco[i]].copy() for i in range(max_rounds)], [int(ch[2 * i]) | (int(ch[2 * i + 1]) << 64) for i in range(max_rounds)],
                claims, com[:max_rounds - 1], w, v.reshape(3, max_rounds, 4))

    def free(self):
        if self.h:
            lib.atlas_accumulator_free(self.h)
            self.h = None
