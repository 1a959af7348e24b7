"""jolt-atlas_amd — Python plumbing over libatlas_hip.so (the C-ABI in include/atlas_hip.h).

This is test/bench plumbing, not the product: the product is the shared library, which a
Rust `extern "C"` block binds directly.  The classes mirror the
reference's names so the parity tests read like the reference's own:

  Blake2bTranscript      joltworks/src/transcripts/blake2b.rs
  MultilinearPolynomial  joltworks/src/poly/multilinear_polynomial.rs:22-35
  EinsumDotProver        jolt-atlas-core/src/onnx_proof/ops/einsum/dot.rs:255-375
  Sumcheck.prove         joltworks/src/subprotocols/sumcheck.rs:565-599

There is no CPU fallback: importing works anywhere (so the symbol table can be checked),
but every compute call raises AtlasError when the HIP device or the library is missing.
Fr values are numpy uint64 arrays (..., 4): the ark_bn254::Fr Montgomery limb image.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libatlas_hip.so")

HIGH_TO_LOW, LOW_TO_HIGH = 0, 1
EQ_NONE, EQ_HIGH, EQ_LOW = 0, 1, 2


class AtlasError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise AtlasError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    return C.CDLL(LIB_PATH)


lib = _load()

u64p = C.POINTER(C.c_uint64)


class U128(C.Structure):
    _fields_ = [("lo", C.c_uint64), ("hi", C.c_uint64)]


class TranscriptState(C.Structure):
    _fields_ = [("state", C.c_uint8 * 32), ("n_rounds", C.c_uint32), ("pad_", C.c_uint32 * 3)]


class Timing(C.Structure):
    _fields_ = [("total_ms", C.c_double), ("pass_ms", C.c_double), ("fs_ms", C.c_double),
                ("pass_bytes", C.c_uint64), ("n_pass", C.c_uint32), ("n_fs", C.c_uint32)]


lib.atlas_last_error.restype = C.c_char_p
for _name in ("atlas_poly_upload_fr", "atlas_poly_upload_i32", "atlas_poly_wrap_device_fr", "atlas_poly_len",
              "atlas_poly_download", "atlas_poly_clone", "atlas_poly_free", "atlas_poly_bind",
              "atlas_poly_final_claim", "atlas_dot_prover_new", "atlas_dot_prover_free",
              "atlas_dot_compute_message", "atlas_dot_input_claim", "atlas_dot_ingest_challenge", "atlas_dot_final_claims",
              "atlas_sumcheck_prove_dot"):
    getattr(lib, _name).restype = C.c_int


def _check(rc):
    if rc != 0:
        raise AtlasError(f"atlas error {rc}: {lib.atlas_last_error().decode()}")


def _fr(a):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    assert a.shape[-1] == 4
    return a


def _p(a):
    return a.ctypes.data_as(u64p)


def init(device=0):
    _check(lib.atlas_init(C.c_int(device)))


def init_thread(device=0):
    """one process, N devices: give the CALLING THREAD a runtime of its own on `device` (include/atlas_hip.h: atlas_init_thread)"""
    _check(lib.atlas_init_thread(C.c_int(device)))


def shutdown_thread():
    _check(lib.atlas_shutdown_thread())


def device_count():
    return lib.atlas_device_count()


def device_memory(reset_peak=False):
    """(bytes in use, peak bytes since the last reset) of the calling thread's runtime (atlas_device_memory_stats)"""
    a, b = C.c_size_t(), C.c_size_t()
    _check(lib.atlas_device_memory_stats(C.byref(a), C.byref(b), C.c_int(1 if reset_peak else 0)))
    return a.value, b.value


def set_timeouts(device_wait_s=0.0, host_wait_s=0.0, board_wait_s=0.0):
    """atlas_set_timeouts: how long a launch waits for a challenge / the host for mail / a rank for the board, in seconds (0 keeps a setting)"""
    lib.atlas_set_timeouts.argtypes = [C.c_double, C.c_double, C.c_double]
    _check(lib.atlas_set_timeouts(device_wait_s, host_wait_s, board_wait_s))


def get_timeouts():
    d, h, b = C.c_double(), C.c_double(), C.c_double()
    _check(lib.atlas_get_timeouts(C.byref(d), C.byref(h), C.byref(b)))
    return d.value, h.value, b.value


def set_challenge_mode(mode):
    _check(lib.atlas_set_challenge_mode(C.c_int(mode)))


FS_HOST, FS_DEVICE = 0, 1


def set_fs_mode(mode):
    """ATLAS_FS_HOST (default): transcript on the host over the round channel; ATLAS_FS_DEVICE: on one wavefront."""
    _check(lib.atlas_set_fs_mode(C.c_int(mode)))


def get_fs_mode():
    return int(lib.atlas_get_fs_mode())


def measure_mad_peak():
    """v_mad_u64_u32 per second over the whole chip, measured now."""
    v = C.c_double()
    _check(lib.atlas_measure_mad_peak(C.byref(v)))
    return v.value


def set_timing(on):
    _check(lib.atlas_set_timing(C.c_int(1 if on else 0)))


def last_timing():
    t = Timing()
    _check(lib.atlas_last_timing(C.byref(t)))
    return t


def sync():
    _check(lib.atlas_sync())


def challenge_to_fr(c128):
    out = np.zeros(4, dtype=np.uint64)
    u = U128(c128 & ((1 << 64) - 1), c128 >> 64)
    _check(lib.atlas_challenge_to_fr(C.byref(u), _p(out)))
    return out


class Blake2bTranscript:
    """Host-side transcript (Transcript trait, transcripts/transcript.rs:6-28)."""

    def __init__(self, label: bytes):
        self.t = TranscriptState()
        _check(lib.atlas_transcript_new(C.byref(self.t), label, C.c_size_t(len(label))))

    @property
    def state(self):
        return bytes(self.t.state)

    @property
    def n_rounds(self):
        return self.t.n_rounds

    def append_message(self, msg: bytes):
        _check(lib.atlas_transcript_append_message(C.byref(self.t), msg, C.c_size_t(len(msg))))

    def append_bytes(self, b: bytes):
        _check(lib.atlas_transcript_append_bytes(C.byref(self.t), b, C.c_size_t(len(b))))

    def append_u64(self, x):
        _check(lib.atlas_transcript_append_u64(C.byref(self.t), C.c_uint64(x)))

    def append_scalar(self, fr):
        fr = _fr(fr)
        _check(lib.atlas_transcript_append_scalar(C.byref(self.t), _p(fr)))

    def append_scalars(self, frs):
        frs = _fr(frs)
        _check(lib.atlas_transcript_append_scalars(C.byref(self.t), _p(frs), C.c_size_t(len(frs))))

    def challenge_u128(self):
        u = U128()
        _check(lib.atlas_transcript_challenge_u128(C.byref(self.t), C.byref(u)))
        return u.lo | (u.hi << 64)

    def challenge_scalar(self):
        out = np.zeros(4, dtype=np.uint64)
        _check(lib.atlas_transcript_challenge_scalar(C.byref(self.t), _p(out)))
        return out


class MultilinearPolynomial:
    """Device-resident MLE: LargeScalars (Fr) or I32Scalars."""

    def __init__(self, handle):
        self.h = handle

    @classmethod
    def from_fr(cls, arr):
        arr = _fr(arr)
        h = C.c_void_p()
        _check(lib.atlas_poly_upload_fr(_p(arr), C.c_size_t(arr.shape[0]), C.byref(h)))
        return cls(h)

    @classmethod
    def from_i32(cls, arr):
        arr = np.ascontiguousarray(arr, dtype=np.int32)
        h = C.c_void_p()
        _check(lib.atlas_poly_upload_i32(arr.ctypes.data_as(C.POINTER(C.c_int32)), C.c_size_t(arr.shape[0]),
                                         C.byref(h)))
        return cls(h)

    def len(self):
        n = C.c_size_t()
        _check(lib.atlas_poly_len(self.h, C.byref(n)))
        return n.value

    def clone(self):
        h = C.c_void_p()
        _check(lib.atlas_poly_clone(self.h, C.byref(h)))
        return MultilinearPolynomial(h)

    def to_host(self):
        n = self.len()
        out = np.zeros((n, 4), dtype=np.uint64)
        _check(lib.atlas_poly_download(self.h, _p(out), C.c_size_t(n)))
        return out

    def bind_parallel(self, r_u128, order):
        u = U128(r_u128 & ((1 << 64) - 1), r_u128 >> 64)
        _check(lib.atlas_poly_bind(self.h, C.byref(u), C.c_int(order)))

    def final_claim(self):
        out = np.zeros(4, dtype=np.uint64)
        _check(lib.atlas_poly_final_claim(self.h, _p(out)))
        return out

    def free(self):
        if self.h:
            lib.atlas_poly_free(self.h)
            self.h = None


class EinsumDotProver:
    """SumcheckInstanceProver for the einsum contraction (dot.rs:255-375)."""

    def __init__(self, left, right, eq=None, schedule=EQ_NONE, a=0, b=0):
        h = C.c_void_p()
        _check(lib.atlas_dot_prover_new(left.h, right.h, eq.h if eq is not None else None, C.c_int(schedule),
                                        C.c_size_t(a), C.c_size_t(b), C.byref(h)))
        left.h = right.h = None  # ownership moved
        if eq is not None:
            eq.h = None
        self.h = h
        self.deg = 2 if schedule == EQ_NONE else 3

    def input_claim(self):
        out = np.zeros(4, dtype=np.uint64)
        _check(lib.atlas_dot_input_claim(self.h, _p(out)))
        return out

    def compute_message(self, rnd, previous_claim):
        out = np.zeros((4, 4), dtype=np.uint64)
        n = C.c_size_t()
        pc = _fr(previous_claim)
        _check(lib.atlas_dot_compute_message(self.h, C.c_size_t(rnd), _p(pc), _p(out), C.byref(n)))
        return out[:n.value]

    def ingest_challenge(self, r_u128, rnd):
        u = U128(r_u128 & ((1 << 64) - 1), r_u128 >> 64)
        _check(lib.atlas_dot_ingest_challenge(self.h, C.byref(u), C.c_size_t(rnd)))

    def final_claims(self):
        out = np.zeros((3, 4), dtype=np.uint64)
        _check(lib.atlas_dot_final_claims(self.h, _p(out)))
        return out

    def free(self):
        if self.h:
            lib.atlas_dot_prover_free(self.h)
            self.h = None


class Sumcheck:
    @staticmethod
    def prove(prover: EinsumDotProver, input_claim, transcript: Blake2bTranscript, n_rounds):
        """Sumcheck::prove with the transcript resident on the device.
        Returns (compressed_polys (n_rounds, deg, 4), challenges [u128], final_claims (3,4))."""
        deg = prover.deg
        proof = np.zeros((max(n_rounds, 1) * deg, 4), dtype=np.uint64)
        ch = np.zeros(2 * max(n_rounds, 1), dtype=np.uint64)
        fin = np.zeros((3, 4), dtype=np.uint64)
        ic = _fr(input_claim)
        _check(lib.atlas_sumcheck_prove_dot(prover.h, _p(ic), C.byref(transcript.t), _p(proof), _p(ch), _p(fin)))
        chal = [int(ch[2 * i]) | (int(ch[2 * i + 1]) << 64) for i in range(n_rounds)]
        return proof[:n_rounds * deg].reshape(n_rounds, deg, 4), chal, fin


    @staticmethod
    def verify(rows, claim, transcript: Blake2bTranscript, degree_bound):
        """SumcheckInstanceProof::verify (sumcheck.rs:653-686).  rows: (n_rounds, deg, 4) array or a list of (k, 4) compressed
        coefficient arrays.  Returns (final_claim (4,), challenges [u128]); raises AtlasError (code -5) on a rejected proof."""
        comp, nco, stride = _pack_rows(rows)
        n = len(nco)
        ch = np.zeros(2 * max(n, 1), dtype=np.uint64)
        e = np.zeros(4, dtype=np.uint64)
        c = _fr(claim)
        _check(lib.atlas_sumcheck_proof_verify(_p(comp), C.c_size_t(stride), nco.ctypes.data_as(C.c_void_p), C.c_size_t(n),
                                               C.c_size_t(degree_bound), _p(c), C.byref(transcript.t), _p(e), _p(ch)))
        return e, [int(ch[2 * i]) | (int(ch[2 * i + 1]) << 64) for i in range(n)]


def _pack_rows(rows):
    """list of (k_i, 4) coefficient arrays (or an (n, k, 4) array) -> (n, stride, 4) buffer, n_coeffs, stride"""
    rows = [np.ascontiguousarray(r, dtype=np.uint64).reshape(-1, 4) for r in rows]
    stride = max([len(r) for r in rows] + [1])
    comp = np.zeros((max(len(rows), 1), stride, 4), dtype=np.uint64)
    for i, r in enumerate(rows):
        comp[i, :len(r)] = r
    return comp, np.array([len(r) for r in rows], dtype=np.uint32), stride


_FR_MOD = np.array([0x43e1f593f0000001, 0x2833e84879b97091, 0xb85045b68181585d, 0x30644e72e131a029], dtype=np.uint64)


def random_fr(n, seed):
    """n uniform Fr as Montgomery limbs: draw 4 x u64, clear the top two bits, rejection-
    sample < r (SURVEY §8d input recipe; PCG64 stands in for ChaCha20 — these are inputs,
    not protocol outputs).  Any residue < r is a valid Montgomery image."""
    rng = np.random.Generator(np.random.PCG64(seed))
    out = rng.integers(0, 1 << 64, size=(n, 4), dtype=np.uint64)
    out[:, 3] &= np.uint64((1 << 62) - 1)
    while True:
        ge = np.zeros(n, dtype=bool)
        und = np.ones(n, dtype=bool)
        for k in (3, 2, 1, 0):
            ge |= und & (out[:, k] > _FR_MOD[k])
            und &= out[:, k] == _FR_MOD[k]
        bad = np.nonzero(ge | und)[0]
        if bad.size == 0:
            return out
        new = rng.integers(0, 1 << 64, size=(bad.size, 4), dtype=np.uint64)
        new[:, 3] &= np.uint64((1 << 62) - 1)
        out[bad] = new


# ---------------------------------------------------------------- SRS / MSM (HyperKZG)
G1_DTYPE = np.dtype([("x", np.uint64, (4,)), ("y", np.uint64, (4,)), ("infinity", np.uint64)])  # 72 bytes

for _name in ("atlas_srs_upload", "atlas_srs_generate", "atlas_srs_len", "atlas_srs_download", "atlas_srs_free",
              "atlas_msm_fr", "atlas_msm_poly", "atlas_g1_sum_indexed"):
    getattr(lib, _name).restype = C.c_int


class SRS:
    """KZGProverKey::g1_powers resident in HBM (hyperkzg/kzg.rs:107-143)."""

    def __init__(self, handle):
        self.h = handle

    @classmethod
    def upload(cls, bases):
        """bases: numpy array of G1_DTYPE (the 72-byte arkworks G1Affine image)."""
        bases = np.ascontiguousarray(bases, dtype=G1_DTYPE)
        h = C.c_void_p()
        _check(lib.atlas_srs_upload(bases.ctypes.data_as(C.c_void_p), C.c_size_t(len(bases)), C.c_size_t(72),
                                    C.byref(h)))
        return cls(h)

    @classmethod
    def generate(cls, tau_fr, n):
        """bases[i] = tau^(i+1) * G (SRS::setup with a caller-supplied trapdoor)."""
        tau_fr = _fr(tau_fr)
        h = C.c_void_p()
        _check(lib.atlas_srs_generate(_p(tau_fr), C.c_size_t(n), C.byref(h)))
        return cls(h)

    def len(self):
        n = C.c_size_t()
        _check(lib.atlas_srs_len(self.h, C.byref(n)))
        return n.value

    def download(self, offset=0, n=None):
        n = self.len() - offset if n is None else n
        out = np.zeros(n, dtype=G1_DTYPE)
        _check(lib.atlas_srs_download(self.h, C.c_size_t(offset), C.c_size_t(n), out.ctypes.data_as(C.c_void_p)))
        return out

    def precompute(self, n_points=0, window_bits=0):
        """Fixed-base table 2^(c j) * g1_powers[i] for the first n_points powers (atlas_srs_precompute); setup-time work."""
        _check(lib.atlas_srs_precompute(self.h, C.c_size_t(n_points), C.c_uint32(window_bits)))
        return self.table_info()

    def precompute_range(self, first_point, n_points=0, window_bits=0):
        """The fixed-base table over the powers [first_point, first_point + n_points) only (atlas_srs_precompute_range): a rank's share."""
        _check(lib.atlas_srs_precompute_range(self.h, C.c_size_t(first_point), C.c_size_t(n_points), C.c_uint32(window_bits)))
        return self.table_info()

    def table_info(self):
        n, c, lv = C.c_size_t(), C.c_uint32(), C.c_uint32()
        _check(lib.atlas_srs_table_info(self.h, C.byref(n), C.byref(c), C.byref(lv)))
        return {"n_points": n.value, "window_bits": c.value, "levels": lv.value}

    def msm(self, scalars, offset=0):
        """VariableBaseMSM::msm: scalars = (n,4) Fr array or a device MultilinearPolynomial."""
        out = np.zeros(1, dtype=G1_DTYPE)
        if isinstance(scalars, MultilinearPolynomial):
            _check(lib.atlas_msm_poly(self.h, C.c_size_t(offset), scalars.h, out.ctypes.data_as(C.c_void_p)))
        else:
            s = _fr(scalars).reshape(-1, 4)
            _check(lib.atlas_msm_fr(self.h, C.c_size_t(offset), _p(s), C.c_size_t(s.shape[0]),
                                    out.ctypes.data_as(C.c_void_p)))
        return out[0]

    _SMALL_KIND = {"uint8": 0, "uint16": 1, "uint32": 2, "uint64": 3, "int32": 4, "int64": 5}

    def msm_small(self, scalars, offset=0):
        """VariableBaseMSM::msm for the U8/U16/U32/U64/I32/I64Scalars variants (msm/mod.rs:38-181)."""
        s = np.ascontiguousarray(scalars)
        kind = self._SMALL_KIND[s.dtype.name]
        out = np.zeros(1, dtype=G1_DTYPE)
        _check(lib.atlas_msm_small(self.h, C.c_size_t(offset), s.ctypes.data_as(C.c_void_p), C.c_size_t(s.size),
                                   C.c_int(kind), out.ctypes.data_as(C.c_void_p)))
        return out[0]

    def sum_indexed(self, indices):
        """HyperKZG::commit_one_hot: sum of bases[k*T + t] over the non-zero coefficients."""
        idx = np.ascontiguousarray(indices, dtype=np.uint32)
        out = np.zeros(1, dtype=G1_DTYPE)
        _check(lib.atlas_g1_sum_indexed(self.h, idx.ctypes.data_as(C.POINTER(C.c_uint32)), C.c_size_t(len(idx)),
                                        out.ctypes.data_as(C.c_void_p)))
        return out[0]

    def commit_one_hot_batch(self, index_rows, Ks):
        """HyperKZG::batch_commit_one_hot: index_rows[r] = int32 addresses per cycle (negative = None), Ks[r] = K."""
        rows = [np.ascontiguousarray(r, dtype=np.int32) for r in index_rows]
        R = len(rows)
        ptrs = (C.c_void_p * max(R, 1))(*[r.ctypes.data for r in rows])
        K = (C.c_size_t * max(R, 1))(*[int(k) for k in Ks])
        T = (C.c_size_t * max(R, 1))(*[len(r) for r in rows])
        out = np.zeros(max(R, 1), dtype=G1_DTYPE)
        _check(lib.atlas_commit_one_hot_batch(self.h, ptrs, K, T, C.c_size_t(R), out.ctypes.data_as(C.c_void_p)))
        return out[:R]

    def commit_lookup_chunks(self, d_lookups, log_T, log_K, log_k_chunk=4):
        """The d RaD commitments of one lookup; d_lookups: instances.DeviceU64 (or a device address) of 2^log_T u64 lookup indices."""
        d = -(-log_K // log_k_chunk)
        out = np.zeros(d, dtype=G1_DTYPE)
        ptr = d_lookups.ptr if hasattr(d_lookups, "ptr") else C.c_void_p(d_lookups)
        _check(lib.atlas_commit_lookup_chunks(self.h, ptr, C.c_size_t(log_T), C.c_size_t(log_K), C.c_size_t(log_k_chunk),
                                              out.ctypes.data_as(C.c_void_p)))
        return out

    def commit_lookup_chunks_multi(self, families, log_k_chunk=4):
        """The RaD commitments of several lookups in one launch; families: [(DeviceU64 | device address, log_T, log_K)].  Returns one array."""
        class _Fam(C.Structure):
            _fields_ = [("d_lookups", C.c_void_p), ("log_T", C.c_size_t), ("log_K", C.c_size_t)]
        n = len(families)
        arr = (_Fam * n)()
        total = 0
        for i, (dl, log_T, log_K) in enumerate(families):
            ptr = dl.ptr if hasattr(dl, "ptr") else C.c_void_p(dl)
            arr[i].d_lookups = ptr.value if hasattr(ptr, "value") else ptr
            arr[i].log_T = log_T; arr[i].log_K = log_K
            total += -(-log_K // log_k_chunk)
        out = np.zeros(total, dtype=G1_DTYPE)
        _check(lib.atlas_commit_lookup_chunks_multi(self.h, arr, C.c_size_t(n), C.c_size_t(log_k_chunk), out.ctypes.data_as(C.c_void_p)))
        return out

    def commit_batch(self, polys):
        """CommitmentScheme::batch_commit over device-resident polynomials."""
        n = len(polys)
        hs = (C.c_void_p * max(n, 1))(*[p.h for p in polys])
        out = np.zeros(max(n, 1), dtype=G1_DTYPE)
        _check(lib.atlas_commit_batch(self.h, hs, C.c_size_t(n), out.ctypes.data_as(C.c_void_p)))
        return out[:n]

    def free(self):
        if self.h:
            lib.atlas_srs_free(self.h)
            self.h = None


for _name in ("atlas_eq_evals", "atlas_poly_evaluate", "atlas_poly_evaluate_many", "atlas_mul_prover_new", "atlas_mul_prover_free",
              "atlas_mul_input_claim", "atlas_sumcheck_prove_mul"):
    getattr(lib, _name).restype = C.c_int


class EqPolynomial:
    @staticmethod
    def evals(r_fr, scaling=None):
        """EqPolynomial::evals(_with_scaling): device table of 2^n entries, r[0] = MSB."""
        r = _fr(r_fr).reshape(-1, 4)
        h = C.c_void_p()
        sc = _fr(scaling) if scaling is not None else None
        _check(lib.atlas_eq_evals(_p(r) if len(r) else None, C.c_size_t(len(r)), _p(sc) if sc is not None else None,
                                  C.byref(h)))
        return MultilinearPolynomial(h)


def evaluate(poly: MultilinearPolynomial, r_fr):
    """PolynomialEvaluation::evaluate at the point r (Fr array, r[0] = MSB)."""
    r = _fr(r_fr).reshape(-1, 4)
    out = np.zeros(4, dtype=np.uint64)
    _check(lib.atlas_poly_evaluate(poly.h, _p(r) if len(r) else None, C.c_size_t(len(r)), _p(out)))
    return out


def batch_evaluate(polys, r_fr):
    """PolynomialEvaluation::batch_evaluate (multilinear_polynomial.rs:682): every polynomial at the same
    point, the eq tables built once and a single synchronisation; returns (len(polys), 4) u64."""
    r = _fr(r_fr).reshape(-1, 4)
    hs = (C.c_void_p * len(polys))(*[p.h for p in polys])
    out = np.zeros((len(polys), 4), dtype=np.uint64)
    _check(lib.atlas_poly_evaluate_many(hs, C.c_size_t(len(polys)), _p(r) if len(r) else None, C.c_size_t(len(r)), _p(out)))
    return out


class MulProver:
    """SumcheckInstanceProver for element-wise multiplication (ops/mul.rs:125-185)."""

    def __init__(self, left, right, w_fr):
        w = _fr(w_fr).reshape(-1, 4)
        h = C.c_void_p()
        _check(lib.atlas_mul_prover_new(left.h, right.h, _p(w), C.c_size_t(len(w)), C.byref(h)))
        left.h = right.h = None
        self.h = h
        self.n = len(w)

    def input_claim(self):
        out = np.zeros(4, dtype=np.uint64)
        _check(lib.atlas_mul_input_claim(self.h, _p(out)))
        return out

    def prove(self, input_claim, transcript):
        """Sumcheck::prove. Returns (compressed_polys (n,3,4), challenges, final_claims (3,4))."""
        n = self.n
        proof = np.zeros((n * 3, 4), dtype=np.uint64)
        ch = np.zeros(2 * n, dtype=np.uint64)
        fin = np.zeros((3, 4), dtype=np.uint64)
        ic = _fr(input_claim)
        _check(lib.atlas_sumcheck_prove_mul(self.h, _p(ic), C.byref(transcript.t), _p(proof), _p(ch), _p(fin)))
        return proof.reshape(n, 3, 4), [int(ch[2 * i]) | (int(ch[2 * i + 1]) << 64) for i in range(n)], fin

    def compute_message(self, rnd, previous_claim):
        out = np.zeros((4, 4), dtype=np.uint64)
        n = C.c_size_t()
        pc = _fr(previous_claim)
        _check(lib.atlas_mul_compute_message(self.h, C.c_size_t(rnd), _p(pc), _p(out), C.byref(n)))
        return out[:n.value]

    def ingest_challenge(self, r_u128, rnd):
        u = U128(r_u128 & ((1 << 64) - 1), r_u128 >> 64)
        _check(lib.atlas_mul_ingest_challenge(self.h, C.byref(u), C.c_size_t(rnd)))

    def final_claims(self):
        out = np.zeros((3, 4), dtype=np.uint64)
        _check(lib.atlas_mul_final_claims(self.h, _p(out)))
        return out

    def free(self):
        if self.h:
            lib.atlas_mul_prover_free(self.h)
            self.h = None


class BatchedSumcheck:
    """BatchedSumcheck::prove (joltworks/src/subprotocols/sumcheck.rs:30-184)."""

    @staticmethod
    def prove(instances, input_claims, transcript: Blake2bTranscript):
        """instances: EinsumDotProver / MulProver objects, in batch order.
        Returns (rows: list of (k,4) compressed coefficient arrays, challenges [u128])."""
        b = C.c_void_p()
        _check(lib.atlas_batched_new(C.byref(b)))
        try:
            mx, stride = 0, 4
            for inst, c in zip(instances, input_claims):
                ic = _fr(c)
                if isinstance(inst, MulProver):
                    _check(lib.atlas_batched_add_mul(b, inst.h, _p(ic)))
                    lib.atlas_mul_num_rounds.restype = C.c_size_t
                    mx = max(mx, lib.atlas_mul_num_rounds(inst.h))
                elif isinstance(inst, EinsumDotProver):
                    _check(lib.atlas_batched_add_dot(b, inst.h, _p(ic)))
                    lib.atlas_dot_num_rounds.restype = C.c_size_t
                    mx = max(mx, lib.atlas_dot_num_rounds(inst.h))
                else:   # generic atlas_instance_t wrapper (instances.Instance)
                    _check(lib.atlas_batched_add_instance(b, inst.h, _p(ic)))
                    mx = max(mx, inst.num_rounds())
                    stride = max(stride, inst.degree() + 1)
            comp = np.zeros((max(mx, 1), stride, 4), dtype=np.uint64)
            nco = np.zeros(max(mx, 1), dtype=np.uint32)
            ch = np.zeros(2 * max(mx, 1), dtype=np.uint64)
            mr = C.c_size_t()
            _check(lib.atlas_batched_prove(b, C.byref(transcript.t), _p(comp), C.c_size_t(stride),
                                           nco.ctypes.data_as(C.c_void_p), _p(ch), C.byref(mr)))
            rows = [comp[i, :nco[i]].copy() for i in range(mr.value)]
            return rows, [int(ch[2 * i]) | (int(ch[2 * i + 1]) << 64) for i in range(mr.value)]
        finally:
            lib.atlas_batched_free(b)


    @staticmethod
    def verify(rows, input_claims, num_rounds, degrees, transcript: Blake2bTranscript):
        """BatchedSumcheck::verify (sumcheck.rs:187-259) up to the output claim.  Returns (output_claim (4,), challenges
        [u128], batching_coeffs (n, 4)); the caller computes the instances' expected output claims (after their
        cache_openings) and calls BatchedSumcheck.check."""
        comp, nco, stride = _pack_rows(rows)
        n_inst = len(num_rounds)
        ic = np.ascontiguousarray(np.stack([_fr(c) for c in input_claims]))
        nr = (C.c_size_t * n_inst)(*[int(x) for x in num_rounds])
        dg = (C.c_size_t * n_inst)(*[int(x) for x in degrees])
        mr = len(nco)
        ch = np.zeros(2 * max(mr, 1), dtype=np.uint64)
        coeffs = np.zeros((n_inst, 4), dtype=np.uint64)
        out = np.zeros(4, dtype=np.uint64)
        _check(lib.atlas_batched_sumcheck_verify(_p(comp), C.c_size_t(stride), nco.ctypes.data_as(C.c_void_p), C.c_size_t(mr), _p(ic), nr, dg,
                                                 C.c_size_t(n_inst), C.byref(transcript.t), _p(coeffs), _p(out), _p(ch)))
        return out, [int(ch[2 * i]) | (int(ch[2 * i + 1]) << 64) for i in range(mr)], coeffs

    @staticmethod
    def check(batching_coeffs, expected_output_claims, output_claim):
        """output_claim == sum coeff_i * expected_i, else AtlasError (SumcheckVerificationError)."""
        co = np.ascontiguousarray(batching_coeffs, dtype=np.uint64).reshape(-1, 4)
        ex = np.ascontiguousarray(np.stack([_fr(c) for c in expected_output_claims]))
        oc = _fr(output_claim)
        _check(lib.atlas_batched_sumcheck_check(_p(co), _p(ex), C.c_size_t(len(co)), _p(oc)))


class Shout:
    """Prover-side table builds of the Shout lookup argument (subprotocols/shout.rs)."""

    @staticmethod
    def read_raf_G(lookup_indices, log_K, eq_r: MultilinearPolynomial):
        idx = np.ascontiguousarray(lookup_indices, dtype=np.uint64)
        h = C.c_void_p()
        lib.atlas_shout_read_raf_G.restype = C.c_int
        _check(lib.atlas_shout_read_raf_G(idx.ctypes.data_as(u64p), C.c_size_t(len(idx)), C.c_size_t(log_K), eq_r.h,
                                          C.byref(h)))
        return MultilinearPolynomial(h)

    @staticmethod
    def ra_evals(lookup_indices, log_K, log_k_chunk, eq_r_cycle: MultilinearPolynomial):
        idx = np.ascontiguousarray(lookup_indices, dtype=np.uint64)
        h = C.c_void_p()
        lib.atlas_shout_ra_evals.restype = C.c_int
        _check(lib.atlas_shout_ra_evals(idx.ctypes.data_as(u64p), C.c_size_t(len(idx)), C.c_size_t(log_K),
                                        C.c_size_t(log_k_chunk), eq_r_cycle.h, C.byref(h)))
        return MultilinearPolynomial(h)

    @staticmethod
    def read_raf_prover(G: MultilinearPolynomial, table, log_K, gamma_fr):
        """ReadRafProver (shout.rs:184-277) as an EinsumDotProver-shaped handle."""
        tab = np.ascontiguousarray(table, dtype=np.int32)
        gm = _fr(gamma_fr)
        h = C.c_void_p()
        lib.atlas_shout_read_raf_prover_new.restype = C.c_int
        _check(lib.atlas_shout_read_raf_prover_new(G.h, tab.ctypes.data_as(C.POINTER(C.c_int32)), C.c_size_t(log_K),
                                                   _p(gm), C.byref(h)))
        G.h = None
        p = EinsumDotProver.__new__(EinsumDotProver)
        p.h = h
        p.deg = 2
        return p


class TensorI32:
    """A row-major Tensor<i32> resident in HBM (einsum operand)."""

    def __init__(self, arr):
        arr = np.ascontiguousarray(arr, dtype=np.int32)
        self.shape = arr.shape
        self.d = C.POINTER(C.c_int32)()
        lib.atlas_i32_upload.restype = C.c_int
        _check(lib.atlas_i32_upload(arr.ctypes.data_as(C.POINTER(C.c_int32)), C.c_size_t(arr.size), C.byref(self.d)))

    def free(self):
        if self.d:
            lib.atlas_i32_free(self.d)
            self.d = None


def fold_rows(t: TensorI32, eq: MultilinearPolynomial):
    """out[j] = sum_h M[j, h] * eq[h]  (mk_kn_mn.rs:66-73, the `right` fold)."""
    rows, cols = t.shape
    h = C.c_void_p()
    lib.atlas_fold_i32_rows.restype = C.c_int
    _check(lib.atlas_fold_i32_rows(t.d, C.c_size_t(rows), C.c_size_t(cols), eq.h, C.byref(h)))
    return MultilinearPolynomial(h)


def fold_cols(t: TensorI32, eq: MultilinearPolynomial):
    """out[j] = sum_i M[i, j] * eq[i]  (mk_kn_mn.rs:57-65, the `left` fold)."""
    rows, cols = t.shape
    h = C.c_void_p()
    lib.atlas_fold_i32_cols.restype = C.c_int
    _check(lib.atlas_fold_i32_cols(t.d, C.c_size_t(rows), C.c_size_t(cols), eq.h, C.byref(h)))
    return MultilinearPolynomial(h)


G2_DTYPE = np.dtype([("x", np.uint64, (2, 4)), ("y", np.uint64, (2, 4)), ("infinity", np.uint64)])
VK_DTYPE = np.dtype([("g1", G1_DTYPE), ("g2", G2_DTYPE), ("beta_g2", G2_DTYPE)])


def g2_generator():
    out = np.zeros(1, dtype=G2_DTYPE)
    _check(lib.atlas_g2_generator(out.ctypes.data_as(C.c_void_p)))
    return out[0]


def g2_mul(p, k_fr):
    out = np.zeros(1, dtype=G2_DTYPE); pp = np.array([p], dtype=G2_DTYPE); k = _fr(np.asarray(k_fr).reshape(1, 4))
    _check(lib.atlas_g2_mul(pp.ctypes.data_as(C.c_void_p), _p(k), out.ctypes.data_as(C.c_void_p)))
    return out[0]


def pairing_check(g1_points, g2_points):
    """multi_pairing(g1_points, g2_points).is_zero()"""
    p = np.ascontiguousarray(g1_points, dtype=G1_DTYPE); q = np.ascontiguousarray(g2_points, dtype=G2_DTYPE)
    assert len(p) == len(q)
    r = C.c_int()
    _check(lib.atlas_pairing_check(p.ctypes.data_as(C.c_void_p), q.ctypes.data_as(C.c_void_p), C.c_size_t(len(p)), C.byref(r)))
    return bool(r.value)


class HyperKZG:
    """CommitmentScheme arithmetic for HyperKZG (commitment_scheme.rs:11-131)."""

    @staticmethod
    def vk_from_trapdoor(tau_fr, g1_first_power):
        """HyperKZGVerifierKey of an SRS generated from a known trapdoor (g1 = the SRS's first power)."""
        vk = np.zeros(1, dtype=VK_DTYPE); g1 = np.array([g1_first_power], dtype=G1_DTYPE); t = _fr(np.asarray(tau_fr).reshape(1, 4))
        _check(lib.atlas_hyperkzg_vk_from_trapdoor(_p(t), g1.ctypes.data_as(C.c_void_p), vk.ctypes.data_as(C.c_void_p)))
        return vk

    @staticmethod
    def verify(vk, commitment, point_u128, y_fr, com, w, v, transcript):
        """HyperKZG::verify: True = accept, False = ProofVerifyError (the transcript advances like the prover's)."""
        ell = len(point_u128)
        pts = (U128 * ell)(*[U128(c & ((1 << 64) - 1), c >> 64) for c in point_u128])
        cm = np.array([commitment], dtype=G1_DTYPE); y = _fr(np.asarray(y_fr).reshape(1, 4))
        com_ = np.ascontiguousarray(com, dtype=G1_DTYPE) if ell > 1 else np.zeros(1, dtype=G1_DTYPE)
        w_ = np.ascontiguousarray(w, dtype=G1_DTYPE); v_ = np.ascontiguousarray(v, dtype=np.uint64).reshape(3 * ell, 4)
        rc = lib.atlas_hyperkzg_verify(vk.ctypes.data_as(C.c_void_p), cm.ctypes.data_as(C.c_void_p), pts, C.c_size_t(ell), _p(y),
                                       com_.ctypes.data_as(C.c_void_p), w_.ctypes.data_as(C.c_void_p), _p(v_), C.byref(transcript.t))
        if rc == 0:
            return True
        if rc == -5:                                   # ATLAS_EVERIFY
            return False
        _check(rc)

    @staticmethod
    def commit(srs: SRS, poly: MultilinearPolynomial):
        """HyperKZG::commit for a dense polynomial = commit_as_univariate (kzg.rs:285-298)."""
        return srs.msm(poly)

    @staticmethod
    def open(srs: SRS, poly: MultilinearPolynomial, point_u128, transcript: Blake2bTranscript):
        """HyperKZG::open. Returns (com (ell-1,), w (3,), v (3, ell, 4))."""
        ell = len(point_u128)
        pts = (U128 * ell)(*[U128(c & ((1 << 64) - 1), c >> 64) for c in point_u128])
        com = np.zeros(max(ell - 1, 1), dtype=G1_DTYPE)
        w = np.zeros(3, dtype=G1_DTYPE)
        v = np.zeros((3 * ell, 4), dtype=np.uint64)
        lib.atlas_hyperkzg_open.restype = C.c_int
        _check(lib.atlas_hyperkzg_open(srs.h, poly.h, pts, C.c_size_t(ell), C.byref(transcript.t),
                                       com.ctypes.data_as(C.c_void_p), w.ctypes.data_as(C.c_void_p), _p(v)))
        return com[:ell - 1], w, v.reshape(3, ell, 4)
