"""ctypes binding of the oracle's BatchedSumcheck restatement (oracle/sumcheck.c, reference
joltworks/src/subprotocols/sumcheck.rs:30-262).  TEST INFRASTRUCTURE ONLY, like orc.py."""
import ctypes as C

import numpy as np

from . import orc

FR = C.c_uint64 * 4
STRIDE = 24


class Inst(C.Structure):
    _fields_ = [("kind", C.c_int), ("schedule", C.c_int), ("n_vars", C.c_size_t), ("sa", C.c_size_t), ("sb", C.c_size_t),
                ("left", C.c_void_p), ("right", C.c_void_p), ("eq", C.c_void_p), ("w", C.c_void_p),
                ("input_claim", FR), ("final_claims", FR * 3), ("state", C.c_void_p)]


def dot_instance(L, R, claim, eq=None, schedule=0, a=0, b=0):
    return dict(kind=0, L=np.ascontiguousarray(L).copy(), R=np.ascontiguousarray(R).copy(),
                eq=None if eq is None else np.ascontiguousarray(eq).copy(), schedule=schedule, a=a, b=b,
                claim=np.ascontiguousarray(claim, dtype=np.uint64).reshape(4), n=int(len(L)).bit_length() - 1)


def mul_instance(L, R, w, claim):
    return dict(kind=1, L=np.ascontiguousarray(L).copy(), R=np.ascontiguousarray(R).copy(), w=np.ascontiguousarray(w).copy(),
                claim=np.ascontiguousarray(claim, dtype=np.uint64).reshape(4), n=len(w))


def ra_instance(inst, claim):
    """an oracle/ra.c instance (orc_ra.Instance) as a batch member."""
    return dict(kind=inst.kind, inst=inst, n=inst.n_rounds, claim=np.ascontiguousarray(claim, dtype=np.uint64).reshape(4))


def batched_prove(instances, t):
    """Returns (rows: list of (k,4) arrays of compressed coefficients, challenges, finals: list of (3,4))."""
    n = len(instances)
    arr = (Inst * n)()
    for i, d in enumerate(instances):
        I = arr[i]
        I.kind = d["kind"]; I.n_vars = d["n"]
        if d["kind"] >= 2:
            I.state = C.addressof(d["inst"].st)
        else:
            I.left = d["L"].ctypes.data; I.right = d["R"].ctypes.data
        if d["kind"] >= 2:
            pass
        elif d["kind"] == 0:
            I.schedule = d["schedule"]; I.sa = d["a"]; I.sb = d["b"]
            I.eq = d["eq"].ctypes.data if d["eq"] is not None else None
        else:
            I.w = d["w"].ctypes.data
        for k in range(4):
            I.input_claim[k] = int(d["claim"][k])
    max_rounds = max(d["n"] for d in instances)
    comp = orc.fr_array(STRIDE * max_rounds); nco = np.zeros(max_rounds, dtype=np.uint32)
    ch = np.zeros(2 * max_rounds, dtype=np.uint64); mr = C.c_size_t(0)
    orc.lib.orc_batched_prove.restype = C.c_int
    rc = orc.lib.orc_batched_prove(arr, C.c_size_t(n), C.byref(t), orc._p(comp), C.c_size_t(STRIDE),
                                   nco.ctypes.data_as(C.c_void_p), orc._p(ch), C.byref(mr))
    assert rc == 0 and mr.value == max_rounds
    comp = comp.reshape(max_rounds, STRIDE, 4)
    rows = [comp[i, :nco[i]].copy() for i in range(max_rounds)]
    finals = [np.array([[arr[i].final_claims[j][k] for k in range(4)] for j in range(3)], dtype=np.uint64) for i in range(n)]
    return rows, orc._u128_list(ch, max_rounds), finals


def batched_verify(rows, input_claims, rounds, t):
    """BatchedSumcheck::verify up to the expected-output check. Returns (e (4,), coeffs (n,4), challenges)."""
    max_rounds = len(rows); n = len(rounds)
    comp = orc.fr_array(STRIDE * max_rounds).reshape(max_rounds, STRIDE, 4); nco = np.zeros(max_rounds, dtype=np.uint32)
    for i, r in enumerate(rows):
        comp[i, :len(r)] = r; nco[i] = len(r)
    comp = np.ascontiguousarray(comp)
    claims = np.ascontiguousarray(np.stack([np.asarray(c, dtype=np.uint64).reshape(4) for c in input_claims]))
    rds = np.asarray(rounds, dtype=np.uint64)
    e = orc.fr_array(1); co = orc.fr_array(n); ch = np.zeros(2 * max_rounds, dtype=np.uint64)
    orc.lib.orc_batched_verify.restype = C.c_int
    rc = orc.lib.orc_batched_verify(orc._p(comp), C.c_size_t(STRIDE), nco.ctypes.data_as(C.c_void_p), C.c_size_t(max_rounds), orc._p(claims),
                                    rds.ctypes.data_as(C.c_void_p), C.c_size_t(n), C.byref(t), orc._p(e), orc._p(co), orc._p(ch))
    assert rc == 0
    return e.reshape(4), co.reshape(n, 4), orc._u128_list(ch, max_rounds)


_KIND = {np.dtype(np.uint8): 0, np.dtype(np.uint16): 1, np.dtype(np.uint32): 2, np.dtype(np.uint64): 3,
         np.dtype(np.int32): 4, np.dtype(np.int64): 5}


def msm_small(bases, scalars):
    """VariableBaseMSM::msm on a narrow-scalar polynomial (msm/mod.rs:38-181)."""
    bases = np.ascontiguousarray(bases, dtype=orc.G1_DTYPE)
    scalars = np.ascontiguousarray(scalars)
    out = np.zeros(1, dtype=orc.G1_DTYPE)
    orc.lib.orc_msm_small(bases.ctypes.data_as(C.c_void_p), scalars.ctypes.data_as(C.c_void_p), C.c_size_t(len(scalars)),
                          C.c_int(_KIND[scalars.dtype]), out.ctypes.data_as(C.c_void_p))
    return out[0]


def rlc_build(dense, onehot):
    """build_materialized_rlc.  dense: list of (array (len,4) Fr | int32 array, coeff (4,));
    onehot: list of (k int32 array of T entries (negative = None), K, coeff (4,))."""
    nd, no = len(dense), len(onehot)
    joint_len = max([len(a) for a, _ in dense] + [K * len(k) for k, K, _ in onehot])
    keep = []
    fr_ptrs = (C.c_void_p * max(nd, 1))(); i32_ptrs = (C.c_void_p * max(nd, 1))()
    lens = (C.c_size_t * max(nd, 1))()
    dco = orc.fr_array(max(nd, 1))
    for j, (a, c) in enumerate(dense):
        a = np.ascontiguousarray(a); keep.append(a)
        if a.dtype == np.int32:
            i32_ptrs[j] = a.ctypes.data; fr_ptrs[j] = None
        else:
            fr_ptrs[j] = a.ctypes.data; i32_ptrs[j] = None
        lens[j] = len(a); dco[j] = c
    k_ptrs = (C.c_void_p * max(no, 1))(); Ts = (C.c_size_t * max(no, 1))()
    oco = orc.fr_array(max(no, 1))
    for j, (k, K, c) in enumerate(onehot):
        k = np.ascontiguousarray(k, dtype=np.int32); keep.append(k)
        k_ptrs[j] = k.ctypes.data; Ts[j] = len(k); oco[j] = c
    joint = orc.fr_array(joint_len)
    orc.lib.orc_rlc_build.restype = None
    orc.lib.orc_rlc_build(fr_ptrs, i32_ptrs, lens, orc._p(dco), C.c_size_t(nd), k_ptrs, Ts, orc._p(oco), C.c_size_t(no),
                          orc._p(joint), C.c_size_t(joint_len))
    return joint
