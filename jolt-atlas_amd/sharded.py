"""One sumcheck instance / one MSM sharded over the GPUs of a node.

One process per GPU, `torch.distributed` for the exchange (backend "nccl" = RCCL over xGMI
on a multi-GPU node; "gloo" works for tests).  The payload per round is world*2 field
elements (64 B per rank): latency-bound, so the collective is a plain all_gather of a tiny
tensor and every rank re-runs the deterministic transcript step — no challenge broadcast.
Plumbing only: all arithmetic is in libatlas_hip.so.
"""
import ctypes as C

import numpy as np

from . import (AtlasError, Blake2bTranscript, EinsumDotProver, MultilinearPolynomial, G1_DTYPE, _check, _fr, _p, lib,
               EQ_NONE)

for _n in ("atlas_dot_shard_begin", "atlas_dot_shard_local_message", "atlas_dot_shard_round", "atlas_dot_shard_local_final",
           "atlas_dot_shard_finish", "atlas_fr_sum", "atlas_g1_sum_affine"):
    getattr(lib, _n).restype = C.c_int


def _all_gather_bytes(dist, arr, device):
    """all_gather of a small numpy uint64 array; returns (world, *arr.shape)."""
    import torch
    t = torch.from_numpy(np.ascontiguousarray(arr).view(np.int64).reshape(-1).copy())
    world = dist.get_world_size()
    if device is not None:
        t = t.to(device)
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    return np.stack([o.cpu().numpy().view(np.uint64).reshape(arr.shape) for o in out])


def strided_shard(full, rank, world):
    """The shard rank `rank` owns: coefficient k*world + rank for every k."""
    return np.ascontiguousarray(full[rank::world])


def fr_sum(values):
    values = _fr(values).reshape(-1, 4)
    out = np.zeros(4, dtype=np.uint64)
    _check(lib.atlas_fr_sum(_p(values), C.c_size_t(len(values)), _p(out)))
    return out


def prove_dot_sharded(dist, left_shard, right_shard, transcript: Blake2bTranscript, device=None, input_claim=None):
    """Sumcheck::prove of sum L*R with the operands sharded over dist's ranks (strided).
    left_shard / right_shard: this rank's (len/world, 4) Fr arrays, or device polynomials.
    Every rank returns the same (compressed_polys, challenges, final_claims, input_claim);
    transcript is updated identically."""
    world = dist.get_world_size()
    if world & (world - 1):
        raise AtlasError("world size must be a power of two")
    pl = left_shard if isinstance(left_shard, MultilinearPolynomial) else MultilinearPolynomial.from_fr(left_shard)
    pr = right_shard if isinstance(right_shard, MultilinearPolynomial) else MultilinearPolynomial.from_fr(right_shard)
    n_local = pl.len().bit_length() - 1
    if n_local < 1:
        raise AtlasError("each rank needs at least two coefficients per operand")
    n_total = n_local + world.bit_length() - 1
    prover = EinsumDotProver(pl, pr, None, EQ_NONE, 0, 0)
    if input_claim is None:
        local = prover.input_claim()
        input_claim = fr_sum(_all_gather_bytes(dist, local, device))
    ic = _fr(input_claim)
    _check(lib.atlas_dot_shard_begin(prover.h, _p(ic), C.byref(transcript.t)))
    part = np.zeros((2, 4), dtype=np.uint64)
    for _ in range(n_local):
        _check(lib.atlas_dot_shard_local_message(prover.h, _p(part)))
        gathered = np.ascontiguousarray(_all_gather_bytes(dist, part, device))       # (world, 2, 4)
        _check(lib.atlas_dot_shard_round(prover.h, _p(gathered), C.c_size_t(world)))
    _check(lib.atlas_dot_shard_local_final(prover.h, _p(part)))
    gathered = np.ascontiguousarray(_all_gather_bytes(dist, part, device))
    proof = np.zeros((n_total * 2, 4), dtype=np.uint64)
    ch = np.zeros(2 * n_total, dtype=np.uint64)
    fin = np.zeros((3, 4), dtype=np.uint64)
    _check(lib.atlas_dot_shard_finish(prover.h, _p(gathered), C.c_size_t(world), C.byref(transcript.t), _p(proof), _p(ch),
                                      _p(fin)))
    prover.free()
    chal = [int(ch[2 * i]) | (int(ch[2 * i + 1]) << 64) for i in range(n_total)]
    return proof.reshape(n_total, 2, 4), chal, fin, input_claim


def msm_sharded(dist, srs_slice, scalars_slice, device=None):
    """VariableBaseMSM::msm with points and scalars split by contiguous range over the ranks:
    each rank runs Pippenger on its slice, the world partial points are gathered and summed."""
    part = np.zeros(1, dtype=G1_DTYPE)
    part[0] = srs_slice.msm(scalars_slice)
    raw = part.view(np.uint64).reshape(-1)
    gathered = _all_gather_bytes(dist, raw, device)                                   # (world, 9)
    pts = np.ascontiguousarray(gathered.reshape(-1)).view(G1_DTYPE)
    out = np.zeros(1, dtype=G1_DTYPE)
    _check(lib.atlas_g1_sum_affine(pts.ctypes.data_as(C.c_void_p), C.c_size_t(len(pts)), out.ctypes.data_as(C.c_void_p)))
    return out[0]


# ---- the same without a collective: shared-memory board + round channel, one library call per rank -------------------
class ShardGroup:
    """atlas_shard_group_t: the ranks of one node exchanging 64-byte records through POSIX shared memory."""

    def __init__(self, name, world, rank):
        self.h = C.c_void_p()
        self.world, self.rank = world, rank
        _check(lib.atlas_shard_group_open(name.encode(), C.c_int(world), C.c_int(rank), C.byref(self.h)))

    def allgather(self, arr):
        a = np.ascontiguousarray(arr)
        out = np.zeros((self.world,) + a.shape, dtype=a.dtype)
        _check(lib.atlas_shard_allgather(self.h, a.ctypes.data_as(C.c_void_p), C.c_size_t(a.nbytes), out.ctypes.data_as(C.c_void_p)))
        return out

    def fail_exchange(self, code):
        """this rank gives up on the sharded call in progress: marks the exchange the others are about to make (atlas_shard_fail_exchange)"""
        _check(lib.atlas_shard_fail_exchange(self.h, C.c_int(code)))

    def remote_failed(self):
        """after an exchange raised: (rank that had given up, its code); rank -1 = nobody did (a timeout)"""
        r, c = C.c_int(-1), C.c_int(0)
        _check(lib.atlas_shard_remote_failed(self.h, C.byref(r), C.byref(c)))
        return r.value, c.value

    def close(self):
        if self.h:
            lib.atlas_shard_group_close(self.h)
            self.h = None


def prove_dot_sharded_shm(group: ShardGroup, left_shard, right_shard, transcript: Blake2bTranscript, input_claim=None):
    """Sumcheck::prove of sum L*R, operands sharded (strided) over the group's ranks; one library call per rank
    (atlas_sumcheck_prove_dot_sharded).  Returns (compressed_polys, challenges, final_claims, input_claim)."""
    pl = left_shard if isinstance(left_shard, MultilinearPolynomial) else MultilinearPolynomial.from_fr(left_shard)
    pr = right_shard if isinstance(right_shard, MultilinearPolynomial) else MultilinearPolynomial.from_fr(right_shard)
    n_local = pl.len().bit_length() - 1
    n_total = n_local + group.world.bit_length() - 1
    prover = EinsumDotProver(pl, pr, None, EQ_NONE, 0, 0)
    if input_claim is None:
        input_claim = fr_sum(group.allgather(prover.input_claim()))
    ic = _fr(input_claim)
    proof = np.zeros((n_total * 2, 4), dtype=np.uint64)
    ch = np.zeros(2 * n_total, dtype=np.uint64)
    fin = np.zeros((3, 4), dtype=np.uint64)
    _check(lib.atlas_sumcheck_prove_dot_sharded(prover.h, group.h, _p(ic), C.byref(transcript.t), _p(proof), _p(ch), _p(fin)))
    prover.free()
    return proof.reshape(n_total, 2, 4), [int(ch[2 * i]) | (int(ch[2 * i + 1]) << 64) for i in range(n_total)], fin, input_claim


def prove_elementwise_sharded_shm(group: ShardGroup, op, operand_blocks, r_node_output, transcript: Blake2bTranscript, input_claim, constants=None):
    """One element-wise operator sumcheck (instances.elementwise: Mul, Add, Sub, Square, Iff, ...) sharded by contiguous blocks
    (atlas_elementwise_prove_sharded).  operand_blocks: this rank's block of every operand (device polynomials of 2^(n - log2 world)
    coefficients); r_node_output: the whole opening point (n coordinates).  Returns (rows of compressed coefficients, challenges, final claims)."""
    from . import instances, U128
    rn = np.ascontiguousarray(r_node_output, dtype=np.uint64).reshape(-1, 4)
    lw = group.world.bit_length() - 1
    n = len(rn)
    inst = instances.elementwise(op, operand_blocks, rn[lw:], constants)
    stride = inst.degree() + 1
    comp = np.zeros((n, stride, 4), dtype=np.uint64)
    nco = np.zeros(n, dtype=np.uint32); ch = np.zeros(2 * n, dtype=np.uint64)
    fin = np.zeros((16, 4), dtype=np.uint64); nf = C.c_size_t()
    ic = _fr(input_claim); rh = np.ascontiguousarray(rn[:lw]) if lw else np.zeros((1, 4), dtype=np.uint64)
    lib.atlas_elementwise_prove_sharded.restype = C.c_int
    _check(lib.atlas_elementwise_prove_sharded(inst.h, group.h, _p(rh), _p(ic), C.byref(transcript.t), _p(comp), C.c_size_t(stride),
                                               nco.ctypes.data_as(C.c_void_p), _p(ch), _p(fin), C.c_size_t(16), C.byref(nf)))
    inst.free()
    return [comp[i, :nco[i]].copy() for i in range(n)], [int(ch[2 * i]) | (int(ch[2 * i + 1]) << 64) for i in range(n)], fin[:nf.value].copy()


def hyperkzg_open_sharded_shm(group: ShardGroup, srs, poly: MultilinearPolynomial, point_u128, transcript: Blake2bTranscript):
    """HyperKZG::open with its commitment MSMs split by point range over the group's ranks (atlas_hyperkzg_open_sharded): every rank holds
    the whole polynomial and SRS and gets the whole proof.  Returns (com (ell-1,), w (3,), v (3, ell, 4)) as HyperKZG.open does."""
    from . import U128
    ell = len(point_u128)
    pts = (U128 * ell)(*[U128(c & ((1 << 64) - 1), c >> 64) for c in point_u128])
    com = np.zeros(max(ell - 1, 1), dtype=G1_DTYPE)
    w = np.zeros(3, dtype=G1_DTYPE)
    v = np.zeros((3 * ell, 4), dtype=np.uint64)
    lib.atlas_hyperkzg_open_sharded.restype = C.c_int
    _check(lib.atlas_hyperkzg_open_sharded(srs.h, group.h, poly.h, pts, C.c_size_t(ell), C.byref(transcript.t),
                                           com.ctypes.data_as(C.c_void_p), w.ctypes.data_as(C.c_void_p), _p(v)))
    return com[:ell - 1], w, v.reshape(3, ell, 4)


def msm_sharded_shm(group: ShardGroup, srs_slice, scalars_slice, offset=0):
    """Point-range sharded MSM: one partial point per rank through the board, summed on every rank."""
    part = np.zeros(1, dtype=G1_DTYPE)
    part[0] = srs_slice.msm(scalars_slice, offset)
    pts = np.ascontiguousarray(group.allgather(part.view(np.uint64).reshape(-1)).reshape(-1)).view(G1_DTYPE)
    out = np.zeros(1, dtype=G1_DTYPE)
    _check(lib.atlas_g1_sum_affine(pts.ctypes.data_as(C.c_void_p), C.c_size_t(len(pts)), out.ctypes.data_as(C.c_void_p)))
    return out[0]
