"""ctypes binding of liboracle.so — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
Fr arrays are numpy uint64 arrays of shape (n, 4): Montgomery limbs, little-endian,
exactly the memory layout of ark_bn254::Fr.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liboracle.so")


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


if not os.path.exists(_LIB):
    build()
lib = C.CDLL(_LIB)

u64p = C.POINTER(C.c_uint64)
u8p = C.POINTER(C.c_uint8)
i32p = C.POINTER(C.c_int32)


class Transcript(C.Structure):
    _fields_ = [("state", C.c_uint8 * 32), ("n_rounds", C.c_uint32), ("history", u8p),
                ("history_cap", C.c_size_t), ("history_len", C.c_size_t)]

    def state_bytes(self):
        return bytes(self.state)


def _p(a):
    return a.ctypes.data_as(u64p)


def fr_array(n):
    return np.zeros((n, 4), dtype=np.uint64)


def new_transcript(label: bytes, record=0):
    t = Transcript()
    lib.orc_transcript_new(C.byref(t), label)
    hist = None
    if record:
        hist = np.zeros(record * 32, dtype=np.uint8)
        lib.orc_transcript_record(C.byref(t), hist.ctypes.data_as(u8p), C.c_size_t(record))
        t._hist = hist
    return t


def history(t):
    n = min(t.history_len, t.history_cap)
    return [bytes(t._hist[32 * i:32 * i + 32]) for i in range(n)]


def from_ints(vals):
    """canonical Python ints -> Montgomery limb array (via the Python model's math)."""
    from .pymodel import field as F
    out = fr_array(len(vals))
    for i, v in enumerate(vals):
        out[i] = F.limbs64(F.to_mont(v % F.FR))
    return out


def to_ints(arr):
    from .pymodel import field as F
    return [F.from_mont(F.from_limbs64(row)) for row in np.asarray(arr).reshape(-1, 4)]


def random_fr(n, seed):
    """Uniform Fr (as Montgomery limbs): draw 4xu64, clear top 2 bits, rejection-sample
    < r (SURVEY §8d input recipe; numpy PCG64 stands in for ChaCha20, the values are
    inputs, not protocol outputs)."""
    from .pymodel import field as F
    rng = np.random.Generator(np.random.PCG64(seed))
    out = rng.integers(0, 1 << 64, size=(n, 4), dtype=np.uint64)
    out[:, 3] &= np.uint64((1 << 62) - 1)
    mod = np.array(F.limbs64(F.FR), dtype=np.uint64)
    while True:
        # lexicographic compare from the top limb
        ge = np.zeros(n, dtype=bool)
        und = np.ones(n, dtype=bool)
        for k in (3, 2, 1, 0):
            ge |= und & (out[:, k] > mod[k])
            und &= out[:, k] == mod[k]
        ge |= und
        bad = np.nonzero(ge)[0]
        if bad.size == 0:
            return out
        new = rng.integers(0, 1 << 64, size=(bad.size, 4), dtype=np.uint64)
        new[:, 3] &= np.uint64((1 << 62) - 1)
        out[bad] = new


def _u128_list(buf, n):
    return [int(buf[2 * i]) | (int(buf[2 * i + 1]) << 64) for i in range(n)]


lib.orc_num_threads.restype = C.c_int
lib.orc_sumcheck_dot_prove.restype = C.c_int
lib.orc_sumcheck_dot_prove_i32.restype = C.c_int
lib.orc_sumcheck_verify.restype = C.c_int


def dot_claim(L, R, eq=None, schedule=0, a=0, b=0):
    out = fr_array(1)
    lib.orc_dot_claim(_p(L), _p(R), _p(eq) if eq is not None else None, C.c_size_t(len(L)),
                      C.c_int(schedule), C.c_size_t(a), C.c_size_t(b), _p(out))
    return out


def sumcheck_dot_prove(L, R, claim, t, eq=None, schedule=0, a=0, b=0, consume=False):
    """Binds copies of L/R/eq (or the arrays themselves when consume=True, like the
    reference). Returns (proof (rounds,deg,4), challenges [u128], finals (3,4))."""
    n_vars = (len(L)).bit_length() - 1
    deg = 2 if schedule == 0 else 3
    if consume:
        assert L.flags.c_contiguous and R.flags.c_contiguous
        eqc = eq
    else:
        L = np.ascontiguousarray(L).copy(); R = np.ascontiguousarray(R).copy()
        eqc = None if eq is None else np.ascontiguousarray(eq).copy()
    proof = fr_array(n_vars * deg)
    ch = np.zeros(2 * n_vars, dtype=np.uint64)
    fin = fr_array(3)
    rc = lib.orc_sumcheck_dot_prove(_p(L), _p(R), _p(eqc) if eqc is not None else None,
                                    C.c_size_t(n_vars), C.c_int(schedule), C.c_size_t(a),
                                    C.c_size_t(b), _p(claim), C.byref(t), _p(proof), _p(ch), _p(fin))
    assert rc == 0
    return proof.reshape(n_vars, deg, 4), _u128_list(ch, n_vars), fin


def sumcheck_dot_prove_i32(L, R, claim, t):
    n_vars = (len(L)).bit_length() - 1
    L = np.ascontiguousarray(L, dtype=np.int32); R = np.ascontiguousarray(R, dtype=np.int32)
    proof = fr_array(n_vars * 2)
    ch = np.zeros(2 * n_vars, dtype=np.uint64)
    fin = fr_array(3)
    rc = lib.orc_sumcheck_dot_prove_i32(L.ctypes.data_as(i32p), R.ctypes.data_as(i32p),
                                        C.c_size_t(n_vars), _p(claim), C.byref(t), _p(proof),
                                        _p(ch), _p(fin))
    assert rc == 0
    return proof.reshape(n_vars, 2, 4), _u128_list(ch, n_vars), fin


def sumcheck_verify(proof, claim, t):
    n_rounds, deg = proof.shape[0], proof.shape[1]
    proof = np.ascontiguousarray(proof)
    e = fr_array(1)
    ch = np.zeros(2 * n_rounds, dtype=np.uint64)
    lib.orc_sumcheck_verify(_p(proof), C.c_size_t(n_rounds), C.c_size_t(deg), _p(claim),
                            C.byref(t), _p(e), _p(ch))
    return e, _u128_list(ch, n_rounds)


def eq_evals(r_arr):
    n = len(r_arr)
    out = fr_array(1 << n)
    r_arr = np.ascontiguousarray(r_arr)
    lib.orc_eq_evals(_p(r_arr), C.c_size_t(n), None, _p(out))
    return out


def challenge_to_fr(c128):
    out = fr_array(1)
    lo = C.c_uint64(c128 & ((1 << 64) - 1)); hi = C.c_uint64(c128 >> 64)

    class U128(C.Structure):
        _fields_ = [("lo", C.c_uint64), ("hi", C.c_uint64)]
    lib.fr_from_challenge.argtypes = [U128, u64p]
    lib.fr_from_challenge(U128(lo, hi), _p(out))
    return out


G1_DTYPE = np.dtype([("x", np.uint64, (4,)), ("y", np.uint64, (4,)), ("infinity", np.uint64)])  # g1_aff_t


_SRS_CACHE = {}


def srs_powers(tau_fr, n):
    """bases[i] = tau^(i+1) * G as a G1_DTYPE array (orc_srs_powers).  The powers of one tau are prefixes of each other, so the longest
    request per tau is kept and shorter ones are slices of it (a test session asks for the same tau at many sizes)."""
    t = np.ascontiguousarray(tau_fr, dtype=np.uint64).reshape(1, 4)
    key = t.tobytes()
    have = _SRS_CACHE.get(key)
    if have is None or len(have) < n:
        out = np.zeros(n, dtype=G1_DTYPE)
        lib.orc_srs_powers(_p(t), C.c_size_t(n), out.ctypes.data_as(C.c_void_p))
        if n <= (1 << 18):
            _SRS_CACHE[key] = out
        return out.copy()
    return have[:n].copy()


def msm(bases, scalars, naive=False):
    bases = np.ascontiguousarray(bases, dtype=G1_DTYPE)
    scalars = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 4)
    out = np.zeros(1, dtype=G1_DTYPE)
    fn = lib.orc_msm_naive if naive else lib.orc_msm_pippenger
    fn(bases.ctypes.data_as(C.c_void_p), _p(scalars), C.c_size_t(len(scalars)), out.ctypes.data_as(C.c_void_p))
    return out[0]


def g1_mul_generator(k_fr):
    """k * G for a Montgomery Fr scalar."""
    g = np.zeros(1, dtype=G1_DTYPE)
    lib.g1_generator(g.ctypes.data_as(C.c_void_p))
    out = np.zeros(1, dtype=G1_DTYPE)
    k = np.ascontiguousarray(k_fr, dtype=np.uint64).reshape(1, 4)
    lib.g1_mul_fr(g.ctypes.data_as(C.c_void_p), _p(k), out.ctypes.data_as(C.c_void_p))
    return out[0]


def g1_sum_indexed(bases, idx):
    bases = np.ascontiguousarray(bases, dtype=G1_DTYPE)
    idx = np.ascontiguousarray(idx, dtype=np.uint64)
    out = np.zeros(1, dtype=G1_DTYPE)
    lib.orc_g1_sum_indexed(bases.ctypes.data_as(C.c_void_p), idx.ctypes.data_as(u64p), C.c_size_t(len(idx)),
                           out.ctypes.data_as(C.c_void_p))
    return out[0]


def mul_claim(L, R, w):
    out = fr_array(1)
    w = np.ascontiguousarray(w)
    lib.orc_mul_claim(_p(np.ascontiguousarray(L)), _p(np.ascontiguousarray(R)), _p(w), C.c_size_t(len(w)), _p(out))
    return out


def sumcheck_mul_prove(L, R, w, claim, t):
    """MulProver + Sumcheck::prove. Returns (proof (n,3,4), challenges, finals (3,4))."""
    n = len(w)
    L = np.ascontiguousarray(L).copy(); R = np.ascontiguousarray(R).copy(); w = np.ascontiguousarray(w)
    proof = fr_array(3 * n); ch = np.zeros(2 * n, dtype=np.uint64); fin = fr_array(3)
    lib.orc_sumcheck_mul_prove.restype = C.c_int
    rc = lib.orc_sumcheck_mul_prove(_p(L), _p(R), _p(w), C.c_size_t(n), _p(claim), C.byref(t), _p(proof), _p(ch), _p(fin))
    assert rc == 0
    return proof.reshape(n, 3, 4), _u128_list(ch, n), fin


def challenges_to_fr(c128_list):
    return np.stack([challenge_to_fr(c)[0] for c in c128_list]) if c128_list else fr_array(0)


def hyperkzg_open(srs, poly, point_c128, t):
    """Returns (com (ell-1,) G1, w (3,) G1, v (3, ell, 4) Fr)."""
    ell = len(point_c128)
    srs = np.ascontiguousarray(srs, dtype=G1_DTYPE)
    poly = np.ascontiguousarray(poly, dtype=np.uint64)
    pt = np.ascontiguousarray(challenges_to_fr(point_c128))
    com = np.zeros(max(ell - 1, 1), dtype=G1_DTYPE); w = np.zeros(3, dtype=G1_DTYPE); v = fr_array(3 * ell)
    lib.orc_hyperkzg_open(srs.ctypes.data_as(C.c_void_p), _p(poly), C.c_size_t(ell), _p(pt), C.byref(t),
                          com.ctypes.data_as(C.c_void_p), w.ctypes.data_as(C.c_void_p), _p(v))
    return com[:ell - 1], w, v.reshape(3, ell, 4)


def hyperkzg_verify_trapdoor(srs, tau_fr, commitment, point_c128, y_fr, com, w, v, t):
    ell = len(point_c128)
    srs = np.ascontiguousarray(srs, dtype=G1_DTYPE)
    pt = np.ascontiguousarray(challenges_to_fr(point_c128))
    tau = np.ascontiguousarray(tau_fr, dtype=np.uint64).reshape(1, 4)
    y = np.ascontiguousarray(y_fr, dtype=np.uint64).reshape(1, 4)
    cm = np.ascontiguousarray(np.array([commitment], dtype=G1_DTYPE))
    com = np.ascontiguousarray(com, dtype=G1_DTYPE) if len(com) else np.zeros(1, dtype=G1_DTYPE)
    w = np.ascontiguousarray(w, dtype=G1_DTYPE)
    v = np.ascontiguousarray(v, dtype=np.uint64).reshape(-1, 4)
    lib.orc_hyperkzg_verify_trapdoor.restype = C.c_int
    return bool(lib.orc_hyperkzg_verify_trapdoor(
        srs.ctypes.data_as(C.c_void_p), _p(tau), cm.ctypes.data_as(C.c_void_p), C.c_size_t(ell), _p(pt), _p(y),
        com.ctypes.data_as(C.c_void_p), w.ctypes.data_as(C.c_void_p), _p(v), C.byref(t)))


def evaluate(poly, point_fr):
    n = len(point_fr)
    out = fr_array(1)
    lib.orc_evaluate(_p(np.ascontiguousarray(poly)), C.c_size_t(n), _p(np.ascontiguousarray(point_fr)), _p(out))
    return out[0]


def g1_eq(a, b):
    if int(a["infinity"]) or int(b["infinity"]):
        return bool(int(a["infinity"])) == bool(int(b["infinity"]))
    return np.array_equal(a["x"], b["x"]) and np.array_equal(a["y"], b["y"])


def fr_mul_arr(a, b):
    out = fr_array(1)
    a = np.ascontiguousarray(a, dtype=np.uint64).reshape(1, 4); b = np.ascontiguousarray(b, dtype=np.uint64).reshape(1, 4)
    lib.fr_mul(_p(a), _p(b), _p(out))
    return out[0]


def fr_add_arr(a, b):
    out = fr_array(1)
    a = np.ascontiguousarray(a, dtype=np.uint64).reshape(1, 4); b = np.ascontiguousarray(b, dtype=np.uint64).reshape(1, 4)
    lib.fr_add(_p(a), _p(b), _p(out))
    return out[0]


def serialize_proof(proof):
    """ark CanonicalSerialize bytes of SumcheckInstanceProof."""
    n_rounds, deg = proof.shape[0], proof.shape[1]
    out = n_rounds.to_bytes(8, "little")
    buf = (C.c_uint8 * 32)()
    for i in range(n_rounds):
        out += deg.to_bytes(8, "little")
        for k in range(deg):
            row = np.ascontiguousarray(proof[i, k])
            lib.fr_to_bytes_le(_p(row), buf)
            out += bytes(buf)
    return out
