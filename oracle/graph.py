"""TEST INFRASTRUCTURE ONLY (the checker, never the product): ONNXProof::prove for a model-graph description, composed on
the CPU from the oracle's instances exactly as the reference composes them.

  prove                jolt-atlas-core/src/onnx_proof/mod.rs:153-200
  inputs -> transcript mod.rs:90-122
  commit               prover.rs:71-87, witness.rs:136-200 (BTreeMap<CommittedPoly> order), hyperkzg/mod.rs:520-554
  output_claim         prover.rs:89-121
  iop                  prover.rs:127-138; ops/mod.rs:315-349 (NodeEvalReduction then the operator)
  operators            ops/{add,sub,mul,square,cube,and,iff,relu,reshape,moveaxis,broadcast,identity,input,constant}.rs,
                       ops/einsum/{mod,dot,mk_kn_mn,bmk_rhs_mbn,mbk_rhs_bmn,k_nk_n}.rs, fused_rebase.rs, clamp_lookups/mod.rs,
                       op_lookups/mod.rs, joltworks shout.rs:399-466
  reduced openings     prover.rs:141-176, opening_proof.rs:447-532,611-643
  container            proof_serialization.rs:200-224, types.rs:27-129, opening_proof.rs:1167-1333, common/src/lib.rs

The executor is a numpy statement of the tracer's integer semantics (atlas-onnx-tracer/src/ops/*.rs).  Parity unpinned
against a run of the reference (no Rust toolchain here): this pins the device path against a second, independent
composition — same reading of the reference, different code (Python over oracle/*.c)."""
import ctypes as C

import numpy as np

from . import orc, orc_batched as OB, orc_ra as OR

FR = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001
FQ = 0x30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47

# ---- identifiers (variant index = declaration order in common/src/lib.rs; the u8 tag of canonical_serde_enum!)
VP = {n: i for i, n in enumerate(
    "NodeOutput NodeOutputRa CosRa SinRa TrigDownscaleRa SoftmaxSumOutput SoftmaxMaxOutput SoftmaxMaxIndex HammingWeight DivRangeCheckRa "
    "SqrtRangeCheckRa TeleportRangeCheckRa MeanOfSquaresRangeCheckRa DivRemainder SqrtRemainder TeleportQuotient TeleportRemainder TrigDownscaled "
    "SoftmaxExpSum SoftmaxExpQ SoftmaxRemainderRa SoftmaxExpHi SoftmaxExpLo SoftmaxExpRemainder SoftmaxExpRemainderRa SoftmaxZHi SoftmaxZLo "
    "SoftmaxZHiRa SoftmaxZLoRa SoftmaxClampWitness SoftmaxClampRa SoftmaxRecipMultRemainder ClampAcc ClampRa RescaleRemainder RescaleRemainderRa "
    "SymmetricClampRa ActivationClampedOutput ActivationClampRa ActivationSmallRa".split())}
CP = {n: i for i, n in enumerate(
    "NodeOutputRaD CosRaD SinRaD TrigDownscaleRaD DivRangeCheckRaD SqrtDivRangeCheckRaD MeanOfSquaresRangeCheckRaD SqrtRangeCheckRaD "
    "TeleportRangeCheckRaD DivNodeQuotient ScalarConstDivNodeRemainder RsqrtQuotient TeleportNodeQuotient GatherRa GatherRaD SoftmaxRemainderRaD "
    "SoftmaxExpRemainderRaD SoftmaxZHiRaD SoftmaxZLoRaD ClampRaD RescaleRemainderRaD SymmetricClampRaD ActivationClampRaD ActivationSmallRaD "
    "SoftmaxClampRaD".split())}
SC = {n: i for i, n in enumerate("NodeExecution Raf RaVirtualization RamHammingBooleanity RamHammingWeight Booleanity HammingWeight RLC "
                                 "BlindFoldBatchOpening NTEvalShift".split())}
PT = {n: i for i, n in enumerate("Execution NeuralTeleport RaOneHotChecks RaHammingWeight RangeCheck SoftmaxStage1 SoftmaxStage2 SoftmaxStage3 "
                                 "SoftmaxStage4 SumReduction EinsumMatmul RescaleRemainderRaChecks RescaleArith TrigDownscaleRaChecks".split())}


def vp_arity(v):
    return 0 if v == VP["HammingWeight"] else 2 if VP["SoftmaxSumOutput"] <= v <= VP["SoftmaxMaxIndex"] else 1


def cp_arity(v):
    return 1 if CP["DivNodeQuotient"] <= v <= CP["GatherRa"] else 2


def virt(name, a=0, b=0):
    return (0, VP[name], a, b)


def comm(name, a=0, b=0):
    return (1, CP[name], a, b)


def oid(poly, sc, idx=0):
    """OpeningId as a tuple whose Python order is the derived Ord of the Rust struct (Virtual < Committed)."""
    return poly + (SC[sc], idx)


def node_exec(poly, node):
    return oid(poly, "NodeExecution", node)


# ---- small helpers over the oracle
def fr(vals):
    return orc.from_ints([int(v) % FR for v in np.asarray(vals).reshape(-1)])


def fr_fast(vals):
    """small signed integers -> Montgomery Fr through the oracle's own conversion (orc_i32_to_fr)"""
    v = np.ascontiguousarray(vals, dtype=np.int32)
    out = orc.fr_array(len(v))
    orc.lib.orc_i32_to_fr(v.ctypes.data_as(C.c_void_p), C.c_size_t(len(v)), orc._p(out))
    return out


def one():
    return orc.from_ints([1])[0]


def ilog2(x):
    return int(x).bit_length() - 1


class T:
    """Blake2bTranscript over oracle/transcript.c"""

    def __init__(self, label):
        self.t = orc.new_transcript(label)

    def append_scalar(self, x):
        orc.lib.orc_transcript_append_scalar(C.byref(self.t), orc._p(np.ascontiguousarray(x, dtype=np.uint64).reshape(1, 4)))

    def append_scalars(self, xs):
        xs = np.ascontiguousarray(xs, dtype=np.uint64)
        orc.lib.orc_transcript_append_scalars(C.byref(self.t), orc._p(xs), C.c_size_t(len(xs)))

    def append_message(self, m):
        orc.lib.orc_transcript_append_message(C.byref(self.t), m)

    def append_u64(self, v):
        orc.lib.orc_transcript_append_u64(C.byref(self.t), C.c_uint64(v))

    def append_bytes(self, b):
        buf = (C.c_uint8 * len(b)).from_buffer_copy(b)
        orc.lib.orc_transcript_append_bytes(C.byref(self.t), buf, C.c_size_t(len(b)))

    def challenge_scalar(self):
        s = orc.fr_array(1)
        orc.lib.orc_transcript_challenge_scalar(C.byref(self.t), orc._p(s))
        return s[0].copy()

    def challenge_opt(self):
        raw = (C.c_uint64 * 2)(); r = orc.fr_array(1)
        orc.lib.orc_transcript_challenge_optimized(C.byref(self.t), raw, orc._p(r))
        return r[0].copy()

    def challenge_vector_opt(self, n):
        return np.stack([self.challenge_opt() for _ in range(n)]) if n else orc.fr_array(0)

    def state(self):
        return self.t.state_bytes()


def canon_fq(mont):
    return sum(int(x) << (64 * i) for i, x in enumerate(mont)) * pow(1 << 256, -1, FQ) % FQ


def g1_uncompressed(pt):
    """ark serialize_uncompressed of G1Affine: x LE || y LE with the flags on y's last byte."""
    if int(pt["infinity"]):
        return bytes(63) + b"\x40"
    x, y = canon_fq(pt["x"]), canon_fq(pt["y"])
    b = bytearray(x.to_bytes(32, "little") + y.to_bytes(32, "little"))
    if y > FQ - y:
        b[63] |= 0x80
    return bytes(b)


def g1_compressed(pt):
    if int(pt["infinity"]):
        return bytes(31) + b"\x40"
    x, y = canon_fq(pt["x"]), canon_fq(pt["y"])
    b = bytearray(x.to_bytes(32, "little"))
    if y > FQ - y:
        b[31] |= 0x80
    return bytes(b)


def fr_bytes(x):
    return orc.to_ints(np.asarray(x).reshape(1, 4))[0].to_bytes(32, "little")


def u64(v):
    return int(v).to_bytes(8, "little")


def sumcheck_proof_bytes(rows):
    """SumcheckInstanceProof { compressed_polys: Vec<CompressedUniPoly { coeffs_except_linear_term: Vec<F> }> }"""
    out = u64(len(rows))
    for row in rows:
        out += u64(len(row)) + b"".join(fr_bytes(c) for c in row)
    return out


def opening_id_bytes(k):
    committed, var, a, b, sc, idx = k
    out = bytes([0 if committed else 1, var])
    ar = cp_arity(var) if committed else vp_arity(var)
    if ar >= 1:
        out += u64(a)
    if ar >= 2:
        out += u64(b)
    out += bytes([sc])
    if sc in (SC["NodeExecution"], SC["RLC"]):
        out += u64(idx)
    return out


MODEL_SCALE = 14
ACTIVATION_BOUND = MODEL_SCALE + 3
ACTIVATION_TABLE_VARS = ACTIVATION_BOUND + 1
CLAMP_BOUND = 9                  # joltworks/src/lookup_tables/clamp.rs:197-211


def round_half_away(f):
    """f64::round: half away from zero"""
    import math
    return int(math.floor(abs(f) + 0.5)) * (1 if f >= 0 else -1)


def nonlinearity_value(op, v, sc):
    """One element of tensor::ops::nonlinearities::{tanh, erffunc, sigmoid, sin, cos} (atlas-onnx-tracer/src/tensor/ops.rs:3583-3591,
    3737-3745, 3101-3109, 3420-3428, 3312-3320) at multiplier `sc`: round(sc f(v / sc)) in f64.  The tables below are this function at
    sc = 2^14 (2^8 for the trig tables); the reference's doc-test vectors at other multipliers replay through it
    (tests/test_ref_tensor_ops.py)."""
    import math
    x = v / sc
    if op == "Tanh":
        f = sc * math.tanh(x)
    elif op == "Erf":
        f = sc * _erf_cheb(x)
    elif op == "Sigmoid":
        f = sc / (1.0 + math.exp(-x))
    elif op == "Sin":
        f = sc * math.sin(x)
    elif op == "Cos":
        f = sc * math.cos(x)
    else:
        raise ValueError(op)
    return round_half_away(f)


def tanh_table():
    """materialize_signed_activation_table (neural_teleport/utils.rs:67-85) with nonlinearities::tanh (tensor/ops.rs:3583-3591):
    Table[i] = round(2^14 tanh(signed18(i) / 2^14)) in f64, round-half-away-from-zero like f64::round"""
    return activation_table("Tanh")


_ACT = {}


def _erf_cheb(x):
    """erffunc of the tracer (atlas-onnx-tracer/src/tensor/ops.rs:3671-3735): a Chebyshev fit of erfc, the same f64 operations in the same order"""
    import math
    COF = [-1.3026537197817094, 6.419697923564902e-1, 1.9476473204185836e-2, -9.56151478680863e-3, -9.46595344482036e-4, 3.66839497852761e-4,
           4.2523324806907e-5, -2.0278578112534e-5, -1.624290004647e-6, 1.303655835580e-6, 1.5626441722e-8, -8.5238095915e-8, 6.529054439e-9,
           5.059343495e-9, -9.91364156e-10, -2.27365122e-10, 9.6467911e-11, 2.394038e-12, -6.886027e-12, 8.94487e-13, 3.13092e-13, -1.12708e-13,
           3.81e-16, 7.106e-15, -1.523e-15, -9.4e-17, 1.21e-16, -2.8e-17]

    def erfccheb(z):
        d = dd = 0.0
        t = 2.0 / (2.0 + z); ty = 4.0 * t - 2.0
        for j in range(26, 0, -1):
            d, dd = ty * d - dd + COF[j], d
        return t * math.exp(-(z * z) + 0.5 * (COF[0] + ty * d) - dd)
    return 1.0 - erfccheb(x) if x >= 0.0 else erfccheb(-x) - 1.0


def activation_table(op):
    """materialize_signed_activation_table (neural_teleport/utils.rs:67-85) for Tanh / Erf / Sigmoid (ops/tanh.rs, erf.rs, sigmoid.rs:22-32)"""
    if op not in _ACT:
        n = 1 << ACTIVATION_TABLE_VARS
        sc = float(1 << MODEL_SCALE)
        out = np.zeros(n, dtype=np.int64)
        for i in range(n):
            out[i] = nonlinearity_value(op, i - n if i >= n // 2 else i, sc)
        _ACT[op] = out
    return _ACT[op]


TRIG_PERIOD_MODULUS, TRIG_DOWNSCALE_BITS, TRIG_TABLE_VARS = 2470649, 6, 16     # common/src/consts/trig.rs at MODEL_SCALE = 14 (k = 24)
_TRIG = {}


def trig_table(op):
    """SinTable / CosTable::materialize (neural_teleport/sin.rs:26-41): round(2^8 f(i / 2^8)) * 2^6 over 2^16 indices"""
    if op not in _TRIG:
        sc = float(1 << (MODEL_SCALE - TRIG_DOWNSCALE_BITS))
        out = np.zeros(1 << TRIG_TABLE_VARS, dtype=np.int64)
        for i in range(1 << TRIG_TABLE_VARS):
            out[i] = nonlinearity_value(op, i, sc) * (1 << TRIG_DOWNSCALE_BITS)
        _TRIG[op] = out
    return _TRIG[op]


_EXP_LUT = {}


def exp_lut(scale):
    """generate_exp_lut_decomposed (atlas-onnx-tracer/src/ops/softmax.rs:239-269) in f64 with f64::round (half away from zero);
    lut_hi zero-padded to the next power of two (softmax.rs:94-96).  Returns (lut_hi, lut_lo, log2_base)."""
    if scale not in _EXP_LUT:
        import math
        sf = float(scale)
        needed = int(math.ceil(sf * math.log(2.0 * sf))) + 2
        log2_b = int(math.ceil(math.log2(float(needed)) / 2.0))
        base = 1 << log2_b
        hi_size = needed // base + 2

        def rnd(f):
            return max(int(math.floor(abs(f) + 0.5)) * (1 if f >= 0 else -1), 0)
        hi = [rnd(sf * math.exp(-(float(h) * float(base)) / sf)) for h in range(hi_size)]
        lo = [rnd(sf * math.exp(-float(l) / sf)) for l in range(base)]
        hp = 1 << (hi_size - 1).bit_length()
        _EXP_LUT[scale] = (np.array(hi + [0] * (hp - hi_size), dtype=np.int64), np.array(lo, dtype=np.int64), log2_b)
    return _EXP_LUT[scale]


def softmax_trace(x, F, N, S):
    """softmax_last_axis_decomposed (atlas-onnx-tracer/src/ops/softmax.rs:74-214): the output and the SoftmaxLastAxisTrace"""
    hi, lo, lb = exp_lut(S)
    base = 1 << lb
    z_bound = len(hi) * base
    X = x.astype(np.int64).reshape(F, N)
    mx = X.max(axis=1); am = X.argmax(axis=1)                      # first position of the maximum
    z = mx[:, None] - X
    zc = np.minimum(z, z_bound - 1)
    z_hi, z_lo = zc >> lb, zc & (base - 1)
    e_hi, e_lo = hi[z_hi], lo[z_lo]
    prod = e_hi * e_lo
    exp_q = prod // S; r_exp = prod - exp_q * S
    assert ((r_exp >= 0) & (r_exp < S)).all()
    ssum = exp_q.sum(axis=1)
    inv = (S * S) // ssum
    p = exp_q * inv[:, None]
    sq = p // S; R = p - sq * S
    assert ((R >= 0) & (R < S)).all()
    f = lambda a: a.reshape(-1).astype(np.int32)
    return f(sq), dict(max_k=f(mx), argmax_k=f(am), z=f(z), exp_q=f(exp_q), exp_sum_q=f(ssum), inv_sum=f(inv), R=f(R), z_hi=f(z_hi), z_lo=f(z_lo),
                       exp_hi=f(e_hi), exp_lo=f(e_lo), r_exp=f(r_exp), F=F, N=N, S=S, log2_base=lb, lut_hi=hi.astype(np.int32), lut_lo=lo.astype(np.int32))


# ---- the integer semantics of the tracer (atlas-onnx-tracer/src/ops/*.rs)
def clamp_i32(a):
    return np.clip(a, -(1 << 31), (1 << 31) - 1).astype(np.int32)


EINSUM_NP = {"mk,kn->mn": "mk,kn->mn", "bmk,bkn->mbn": "bmk,bkn->mbn", "bmk,kbn->mbn": "bmk,kbn->mbn", "mbk,bnk->bmn": "mbk,bnk->bmn",
             "mbk,nbk->bmn": "mbk,nbk->bmn", "k,nk->n": "k,nk->n"}


def einsum_operand_shapes(nd):
    s = nd["shape"]
    lay = nd["layout"]
    if lay == "mk,kn->mn":
        m, k, n = s; return (m, k), (k, n)
    if lay == "k,nk->n":
        k, n = s; return (k,), (n, k)
    b, m, k, n = s
    return {"bmk,bkn->mbn": ((b, m, k), (b, k, n)), "bmk,kbn->mbn": ((b, m, k), (k, b, n)), "mbk,bnk->bmn": ((m, b, k), (b, n, k)),
            "mbk,nbk->bmn": ((m, b, k), (n, b, k))}[lay]


def rebase_bits(nd):
    return {"Einsum": nd.get("scale"), "Mul": nd.get("scale"), "Square": nd.get("scale"), "Cube": 2 * nd.get("scale", 0)}.get(nd["op"])


def accumulate(nd, ins):
    """the i64 accumulation of a fused-rescale node (einsum_acc_i64, mul_acc_i64, cube_acc_i64)"""
    op = nd["op"]
    a = ins[0].astype(np.int64)
    if op == "Einsum":
        ls, rs = einsum_operand_shapes(nd)
        return np.einsum(EINSUM_NP[nd["layout"]], a.reshape(ls), ins[1].astype(np.int64).reshape(rs)).reshape(-1)
    if op == "Mul":
        return a * ins[1].astype(np.int64)
    if op == "Square":
        return a * a
    if op == "Cube":
        return a * a * a
    raise ValueError(op)


def execute(nodes, inputs):
    """Model::trace: node idx -> flat int32 output (every dimension is a power of two, so padded == raw)."""
    out, wit = {}, {}
    it = iter(inputs)
    for nd in nodes:
        op, ins = nd["op"], [out[i] for i in nd["inputs"]]
        if op == "Input":
            o = np.ascontiguousarray(next(it), dtype=np.int32).reshape(-1)
        elif op == "Constant":
            o = np.ascontiguousarray(nd["data"], dtype=np.int32).reshape(-1)
        elif op in ("Identity", "Reshape"):
            o = ins[0].copy()
        elif op in ("Add", "Sub"):
            acc = ins[0].astype(np.int64) + ins[1].astype(np.int64) if op == "Add" else ins[0].astype(np.int64) - ins[1].astype(np.int64)
            wit[nd["idx"]] = dict(acc=acc)
            o = clamp_i32(acc)
        elif op in ("Einsum", "Mul", "Square", "Cube"):
            S = rebase_bits(nd)
            acc = accumulate(nd, ins)
            q = acc >> S                                       # div_euclid by 2^S
            r = acc - (q << S)                                 # rem_euclid
            assert ((r >= 0) & (r < (1 << S))).all()
            wit[nd["idx"]] = dict(quot=q, rem=r.astype(np.int32), S=S)
            o = clamp_i32(q)
        elif op == "And":
            o = (ins[0] * ins[1]).astype(np.int32)
        elif op == "Iff":
            o = np.where(ins[0] != 0, ins[1], ins[2]).astype(np.int32)
        elif op == "ReLU":
            o = np.maximum(ins[0], 0).astype(np.int32)
        elif op == "Neg":
            o = (0 - ins[0].astype(np.int64)).astype(np.int32)
        elif op == "IsNan":
            o = np.zeros(int(np.prod(nd["dims"])), dtype=np.int32)
        elif op == "Clamp":                                      # nonlinearities::clamp (tensor/ops.rs:3216-3220), bound_log = CLAMP_BOUND
            assert nd["bound_log"] == CLAMP_BOUND
            o = np.clip(ins[0], -(1 << CLAMP_BOUND), (1 << CLAMP_BOUND) - 1).astype(np.int32)
        elif op == "MoveAxis":
            idims = next(n for n in nodes if n["idx"] == nd["inputs"][0])["dims"]
            o = np.moveaxis(ins[0].reshape(idims), nd["source"], nd["destination"]).reshape(-1).copy()
        elif op == "Broadcast":
            idims = next(n for n in nodes if n["idx"] == nd["inputs"][0])["dims"]
            o = np.broadcast_to(ins[0].reshape(idims), nd["dims"]).reshape(-1).copy()
        elif op == "Slice":
            idims = next(n for n in nodes if n["idx"] == nd["inputs"][0])["dims"]
            sl = [slice(None)] * len(idims); sl[nd["axis"]] = slice(nd["start"], nd["end"])
            o = ins[0].reshape(idims)[tuple(sl)].reshape(-1).copy()
        elif op == "Concat":                                     # tensor::ops::concat (atlas-onnx-tracer/src/tensor/ops.rs:2772)
            parts = [ins[k].reshape(next(n for n in nodes if n["idx"] == j)["dims"]) for k, j in enumerate(nd["inputs"])]
            o = np.concatenate(parts, axis=nd["axis"]).reshape(-1).copy()
        elif op == "Sum":
            idims = next(n for n in nodes if n["idx"] == nd["inputs"][0])["dims"]
            acc = ins[0].astype(np.int64).reshape(idims).sum(axis=nd["axes"][0]).reshape(-1)
            wit[nd["idx"]] = dict(acc=acc)
            o = clamp_i32(acc)
        elif op in ("ScalarConstDiv", "Div"):
            b = np.full(len(ins[0]), nd["divisor"], dtype=np.int64) if op == "ScalarConstDiv" else ins[1].astype(np.int64)
            a = ins[0].astype(np.int64)
            q = np.floor_divide(a, b); r = a - q * b             # remainder with the divisor's sign (adjusted_remainder)
            wit[nd["idx"]] = dict(rem=r.astype(np.int32))
            o = q.astype(np.int32)
        elif op == "MeanOfSquares":
            idims = next(n for n in nodes if n["idx"] == nd["inputs"][0])["dims"]
            N = idims[-1]
            acc = (ins[0].astype(np.int64) ** 2).reshape(-1, N).sum(axis=1)
            D = (1 << nd["scale"]) * nd["count"]
            q = np.floor_divide(acc, D); r = acc - q * D
            wit[nd["idx"]] = dict(quot=q, rem=r.astype(np.int32), D=D)
            o = clamp_i32(q)
        elif op == "Rsqrt":
            s3 = 1 << (3 * nd["scale"])
            x = ins[0].astype(np.int64)
            assert (x > 0).all(), "Rsqrt of a non-positive value"
            q = s3 // x; dr = s3 % x
            out_ = np.array([int(np.floor(np.sqrt(float(v)))) for v in q], dtype=np.int64)
            out_ = np.where(out_ * out_ > q, out_ - 1, out_); out_ = np.where((out_ + 1) * (out_ + 1) <= q, out_ + 1, out_)
            wit[nd["idx"]] = dict(quot=q, div_rem=dr.astype(np.int32), sqrt_rem=(q - out_ * out_).astype(np.int32), bound=(2 * out_ + 1).astype(np.int32))
            o = out_.astype(np.int32)
        elif op in ("Tanh", "Erf", "Sigmoid"):
            assert nd["scale"] == MODEL_SCALE
            c = np.clip(ins[0], -(1 << ACTIVATION_BOUND), (1 << ACTIVATION_BOUND) - 1).astype(np.int32)
            k = np.where(c < 0, c.astype(np.int64) + (1 << ACTIVATION_TABLE_VARS), c.astype(np.int64))
            wit[nd["idx"]] = dict(clamped=c, small_idx=k.astype(np.uint64))
            o = activation_table(op)[k].astype(np.int32)
        elif op in ("Sin", "Cos"):                               # eval_trig (atlas-onnx-tracer/src/ops/mod.rs:317-336), compute_division (Euclidean)
            assert nd["scale"] == MODEL_SCALE
            x = ins[0].astype(np.int64)
            q = np.floor_divide(x, TRIG_PERIOD_MODULUS); rem = x - q * TRIG_PERIOD_MODULUS
            down = rem >> TRIG_DOWNSCALE_BITS
            wit[nd["idx"]] = dict(quot=q.astype(np.int32), rem=rem.astype(np.int32), down=down.astype(np.int32))
            o = trig_table(op)[down].astype(np.int32)
        elif op in ("GatherLarge", "GatherSmall"):
            ddims = next(n for n in nodes if n["idx"] == nd["inputs"][0])["dims"]
            o = ins[0].reshape(ddims[0], -1)[ins[1]].reshape(-1).astype(np.int32)
        elif op == "SoftmaxLastAxis":
            assert nd["scale"] == MODEL_SCALE
            N = nd["dims"][-1]
            o, wit[nd["idx"]] = softmax_trace(ins[0], int(np.prod(nd["dims"])) // N, N, 1 << nd["scale"])
        else:
            raise ValueError(f"oracle graph executor: operator {op}")
        assert len(o) == int(np.prod(nd["dims"])), (nd, len(o))
        out[nd["idx"]] = o
    return out, wit


class Prover:
    def __init__(self, nodes, outputs, srs_host):
        self.nodes = {nd["idx"]: nd for nd in nodes}
        self.order = [nd["idx"] for nd in nodes]
        self.outputs = outputs
        self.srs = srs_host
        self.openings = {}          # OpeningId tuple -> (point (n,4), claim (4,))
        self.reduced = {}
        self.committed = {}         # CommittedPoly tuple -> dict(idx rows, log_T, commitment, point, claim)
        self.proofs = {}            # (node, proof type) -> rows
        self.evalred = {}
        self.t = T(b"ONNXProof")

    # ---- accumulator
    def append_virtual(self, key, point, claim):
        self.t.append_scalar(claim)
        self.openings[key] = (np.array(point, dtype=np.uint64).reshape(-1, 4).copy(), np.array(claim, dtype=np.uint64).copy())

    def append_nodeio(self, nd, pos, point, claim):
        self.append_virtual(node_exec(virt("NodeOutput", nd["inputs"][pos]), nd["idx"]), point, claim)

    def append_sparse(self, cp_name, node, chunk, sc, point, claim):
        self.t.append_scalar(claim)
        p = comm(cp_name, node, chunk)
        self.openings[oid(p, sc)] = (np.array(point, dtype=np.uint64).copy(), np.array(claim, dtype=np.uint64).copy())
        if p in self.committed:                                # sumchecks.insert(label, ..): the last append wins
            self.committed[p]["point"], self.committed[p]["claim"] = np.array(point, dtype=np.uint64).copy(), np.array(claim, dtype=np.uint64).copy()

    def append_dense(self, nd, cp_name, point, claim):
        self.t.append_scalar(claim)
        p = comm(cp_name, nd["idx"])
        self.openings[node_exec(p, nd["idx"])] = (np.array(point, dtype=np.uint64).copy(), np.array(claim, dtype=np.uint64).copy())
        self.committed[p]["point"], self.committed[p]["claim"] = np.array(point, dtype=np.uint64).copy(), np.array(claim, dtype=np.uint64).copy()

    def append_advice(self, nd, vp_name, point, claim):
        self.append_virtual(node_exec(virt(vp_name, nd["idx"]), nd["idx"]), point, claim)

    def mle(self, idx):
        return fr(self.trace[idx])

    # ---- witness + commitments
    def lookup_families(self, nd):
        """[(CommittedPoly name, lookup indices (u64), log_K)] in get_committed_polynomials order"""
        op, i = nd["op"], nd["idx"]
        # one element: the is_scalar operators (add, sub, sum, the fused-rescale family, mean_of_squares; Div by its own branch) commit nothing;
        # the lookup operators proper (ReLU, Clamp, the activations, Rsqrt, Sin / Cos) have no such branch in the reference (ops/relu.rs,
        # clamp.rs, tanh.rs, rsqrt.rs, sin.rs ...): their generic flows run over one cycle, with one-hot polynomials of K x 1 coefficients
        # (nor has GatherLarge: ops/gather/large.rs:106-112 commits its d chunks whatever the shape)
        if int(np.prod(nd["dims"])) == 1 and op not in ("ReLU", "Clamp", "Tanh", "Erf", "Sigmoid", "Rsqrt", "Sin", "Cos", "GatherLarge"):
            return []
        if op in ("Add", "Sub"):
            return [("ClampRaD", self.wit[i]["acc"].astype(np.int64).view(np.uint64), 64)]
        if op in ("Einsum", "Mul", "Square", "Cube"):
            w = self.wit[i]
            return [("RescaleRemainderRaD", w["rem"].astype(np.uint64), w["S"]), ("ClampRaD", w["quot"].astype(np.int64).view(np.uint64), 64)]
        if op == "ReLU":
            return [("NodeOutputRaD", self.trace[nd["inputs"][0]].astype(np.uint32).astype(np.uint64), 32)]
        if op in ("Sin", "Cos"):                                             # ops/sin.rs:169-186
            w = self.wit[i]
            tau = np.full(len(w["rem"]), TRIG_PERIOD_MODULUS, dtype=np.int32)
            return [("TrigDownscaleRaD", w["rem"].astype(np.uint32).astype(np.uint64), 32), ("SinRaD" if op == "Sin" else "CosRaD", w["down"].astype(np.uint64), TRIG_TABLE_VARS),
                    ("TeleportRangeCheckRaD", interleave_arr(w["rem"], tau), 64)]
        if op == "Clamp":
            return [("SymmetricClampRaD", self.trace[nd["inputs"][0]].astype(np.uint32).astype(np.uint64), 32)]
        if op == "Sum":
            return [("ClampRaD", self.wit[i]["acc"].astype(np.int64).view(np.uint64), 64)]
        if op in ("Tanh", "Erf", "Sigmoid"):
            return [("ActivationClampRaD", self.trace[nd["inputs"][0]].astype(np.uint32).astype(np.uint64), 32),
                    ("ActivationSmallRaD", self.wit[i]["small_idx"], ACTIVATION_TABLE_VARS)]
        if op == "GatherLarge":
            V = self.nodes[nd["inputs"][0]]["dims"][0]
            return [("GatherRaD", self.trace[nd["inputs"][1]].astype(np.uint32).astype(np.uint64), ilog2(V))]
        if op == "SoftmaxLastAxis":                                # ops/softmax_last_axis/mod.rs:136-159
            w = self.wit[i]
            u = lambda a: a.astype(np.uint32).astype(np.uint64)
            return [("SoftmaxRemainderRaD", u(w["R"]), MODEL_SCALE), ("SoftmaxExpRemainderRaD", u(w["r_exp"]), MODEL_SCALE), ("SoftmaxClampRaD", u(w["z"]), 32),
                    ("SoftmaxZHiRaD", u(w["z_hi"]), ilog2(len(w["lut_hi"]))), ("SoftmaxZLoRaD", u(w["z_lo"]), ilog2(len(w["lut_lo"])))]
        if op == "Div":
            return [("DivRangeCheckRaD", interleave_arr(self.wit[i]["rem"], self.trace[nd["inputs"][1]]), 64)]
        if op == "MeanOfSquares":
            w = self.wit[i]
            return [("ClampRaD", w["quot"].astype(np.int64).view(np.uint64), 64),
                    ("MeanOfSquaresRangeCheckRaD", interleave_arr(w["rem"], np.full(len(w["rem"]), w["D"], dtype=np.int32)), 64)]
        if op == "Rsqrt":
            w = self.wit[i]
            return [("SqrtDivRangeCheckRaD", interleave_arr(w["div_rem"], self.trace[nd["inputs"][0]]), 64),
                    ("SqrtRangeCheckRaD", interleave_arr(w["sqrt_rem"], w["bound"]), 64)]
        return []

    def dense_committed(self, nd):
        """[(CommittedPoly name, coefficients as Fr)]: the dense advice polynomials of a node"""
        op, i = nd["op"], nd["idx"]
        if op == "ScalarConstDiv":
            return [("ScalarConstDivNodeRemainder", fr(self.wit[i]["rem"]))]
        if op == "Div":
            return [("DivNodeQuotient", fr(self.trace[i]))]
        if op == "Rsqrt":
            return [("RsqrtQuotient", fr(self.wit[i]["quot"]))]
        if op in ("Sin", "Cos"):
            return [("TeleportNodeQuotient", fr(self.wit[i]["quot"]))]
        return []

    def commit(self):
        for i in self.order:
            nd = self.nodes[i]
            if nd["op"] == "GatherSmall":                                    # ops/gather/small.rs:119-121: GatherRa, one polynomial over all dict_len addresses
                V = self.nodes[nd["inputs"][0]]["dims"][0]
                row = self.trace[nd["inputs"][1]].astype(np.int64); Tn = len(row)
                self.committed[comm("GatherRa", i)] = dict(row=row.astype(np.int32), log_T=ilog2(Tn), lk=ilog2(V),
                                                           commitment=orc.g1_sum_indexed(self.srs, (row * Tn + np.arange(Tn)).astype(np.uint64)))
            for name, lookups, log_K in self.lookup_families(nd):
                d = -(-log_K // 4)
                Tn = len(lookups)
                for c in range(d):
                    row = ((lookups >> np.uint64(4 * (d - 1 - c))) & np.uint64(15)).astype(np.int64)
                    flat = (row * Tn + np.arange(Tn)).astype(np.uint64)              # one-hot coefficient k * T + t (one_hot_polynomial.rs:104-113)
                    self.committed[comm(name, i, c)] = dict(row=row.astype(np.int32), log_T=ilog2(Tn), commitment=orc.g1_sum_indexed(self.srs, flat))
            # one element: the is_scalar operators have no committed polynomial; ScalarConstDiv keeps its remainder (generic flow) and Div its
            # quotient (ops/div.rs:157-160: `if node.is_scalar() { return polys; }` after DivNodeQuotient)
            if int(np.prod(nd["dims"])) > 1 or nd["op"] in ("ScalarConstDiv", "Div", "Rsqrt", "Sin", "Cos"):
                for name, coeffs in self.dense_committed(nd):
                    self.committed[comm(name, i)] = dict(dense=coeffs, log_T=ilog2(len(coeffs)), commitment=orc.msm(self.srs[:len(coeffs)], coeffs))
        for key in sorted(self.committed):
            self.t.append_bytes(g1_uncompressed(self.committed[key]["commitment"])[::-1])       # append_serializable

    # ---- generic pieces
    def run(self, inst, claim, node, ptype):
        rows, ch = inst.prove(claim, self.t.t)
        self.proofs[(node, PT[ptype])] = rows
        return orc.challenges_to_fr(ch)

    def onehot_checks(self, nd, lookups, log_K, r_cycle, ra_point, ra_claim, cp_name, ptype):
        self.onehot_checks_multi(nd, [(lookups, log_K, r_cycle, ra_point, ra_claim, cp_name)], ptype)

    def onehot_build(self, fams):
        """ra_onehot_provers per family (shout.rs:399-466; draws in order): the [ra, hw, bool] batch members and what cache_openings needs"""
        lkc = 4
        insts, st = [], []
        for lookups, log_K, r_cycle, ra_point, ra_claim, cp_name in fams:
            d = -(-log_K // lkc)
            q = self.t.challenge_scalar()
            gp = [one()]
            for _ in range(1, d):
                gp.append(orc.fr_mul_arr(gp[-1], q))
            gp = np.stack(gp)
            gammas = self.t.challenge_vector_opt(d)
            r_addr = self.t.challenge_vector_opt(lkc)
            Hs = [((lookups >> np.uint64(lkc * (d - 1 - i))) & np.uint64(15)).astype(np.int32) for i in range(d)]
            G = OR.ra_G(Hs, lkc, r_cycle)
            pad = d * lkc - log_K
            chunks = np.concatenate([np.zeros((pad, 4), dtype=np.uint64), ra_point[:log_K]]).reshape(d, lkc, 4)
            r_cyc_ra = np.ascontiguousarray(ra_point[log_K:])
            hw_claim = orc.fr_array(1)[0]
            for x in gp:
                hw_claim = orc.fr_add_arr(hw_claim, x)
            insts += [OB.ra_instance(OR.ra_virtual(Hs, lkc, chunks, r_cyc_ra), ra_claim), OB.ra_instance(OR.hamming(G, lkc, gp), hw_claim),
                      OB.ra_instance(OR.booleanity(G, Hs, lkc, gammas, r_addr, r_cycle), orc.fr_array(1)[0])]
            st.append((d, Hs, G, chunks, r_cycle, cp_name))
        return insts, st

    def onehot_cache(self, node, st, rs):
        """cache_openings of [RaVirtual, HammingWeight, Booleanity] per family; an instance of n rounds sees the LAST n challenges"""
        lkc, mr = 4, len(rs)
        for d, Hs, G, chunks, r_cycle, cp_name in st:
            log_T = len(r_cycle)
            ra_rs = np.ascontiguousarray(rs[mr - log_T:][::-1])
            for i in range(d):                                               # RaVirtual::cache_openings
                F = orc.eq_evals(chunks[i])
                c = orc.evaluate(np.ascontiguousarray(F[Hs[i]]), ra_rs)
                self.append_sparse(cp_name, node, i, "RaVirtualization", np.concatenate([chunks[i], ra_rs]), c)
            hw_rs = np.ascontiguousarray(rs[mr - lkc:][::-1])
            for i in range(d):                                               # HammingWeight::cache_openings
                c = orc.evaluate(G[i], hw_rs)
                self.append_sparse(cp_name, node, i, "HammingWeight", np.concatenate([hw_rs, r_cycle]), c)
            sl = rs[mr - lkc - log_T:]
            ba = np.ascontiguousarray(sl[:lkc][::-1]); bc = np.ascontiguousarray(sl[lkc:][::-1])
            Fb = orc.eq_evals(ba)
            for i in range(d):                                               # Booleanity::cache_openings
                c = orc.evaluate(np.ascontiguousarray(Fb[Hs[i]]), bc)
                self.append_sparse(cp_name, node, i, "Booleanity", np.concatenate([ba, bc]), c)

    def onehot_checks_multi(self, nd, fams, ptype):
        """ra_onehot_provers per family (draws in order), ONE BatchedSumcheck over [ra, hw, bool] x families, cache_openings in order"""
        insts, st = self.onehot_build(fams)
        rows, ch, _ = OB.batched_prove(insts, self.t.t)
        self.proofs[(nd["idx"], PT[ptype])] = rows
        self.onehot_cache(nd["idx"], st, orc.challenges_to_fr(ch))

    def read_raf(self, nd, inst, claim, lookups, log_K, ra_vp, ptype):
        """Sumcheck::prove of a read-raf instance + its ra opening at (address challenges, reversed cycle challenges)"""
        rs = self.run(inst, claim, nd["idx"], ptype)
        ra_point = np.concatenate([rs[:log_K], rs[log_K:][::-1]])
        ra_claim = OR.ra_claim(lookups, log_K, ra_point)
        self.append_virtual(node_exec(virt(ra_vp, nd["idx"]), nd["idx"]), ra_point, ra_claim)
        return ra_point, ra_claim

    def clamp_lookup(self, nd, cidx, r0, acc_claim, out_claim):
        """prove_clamp_lookup after its raf claim (clamp_lookups/mod.rs:264-309)"""
        gamma = self.t.challenge_scalar()
        exec_claim = orc.fr_add_arr(out_claim, orc.fr_mul_arr(gamma, acc_claim))
        ra_point, ra_claim = self.read_raf(nd, OR.ps_clamp(cidx, 64, 31, True, r0, gamma), exec_claim, cidx, 64, "ClampRa", "Execution")
        self.onehot_checks(nd, cidx, 64, r0, ra_point, ra_claim, "ClampRaD", "RaOneHotChecks")

    # ---- stages
    def output_claim(self):
        nd = self.nodes[self.outputs[0]]
        n = ilog2(len(self.trace[nd["idx"]]))
        r = self.t.challenge_vector_opt(n)
        self.append_virtual(node_exec(virt("NodeOutput", nd["idx"]), nd["idx"] + 1), r, orc.evaluate(self.mle(nd["idx"]), r))

    def eval_reduction(self, nd):
        i = nd["idx"]
        lo, hi = node_exec(virt("NodeOutput", i), i), node_exec(virt("NodeOutput", i), 1 << 64)
        ks = sorted(k for k in self.openings if lo <= k <= hi)
        assert ks, f"node {i} has no opening claims"
        pts = np.stack([self.openings[k][0] for k in ks]); cls = np.stack([self.openings[k][1] for k in ks])
        h, r, c = OR.eval_reduction_prove(self.mle(i), pts, cls, self.t.t)
        self.evalred[i] = h
        self.reduced[i] = (r, c)

    def ew_sumcheck(self, nd, ew, n_ops, in_claim, ptype):
        r0, _ = self.reduced[nd["idx"]]
        I = OR.elementwise(ew, [self.mle(j) for j in nd["inputs"][:n_ops]], r0)
        rs = self.run(I, in_claim, nd["idx"], ptype)
        fin = I.finals()
        pt = np.ascontiguousarray(rs[::-1])
        for q in range(n_ops):
            self.append_nodeio(nd, q, pt, fin[q])

    def einsum_matmul(self, nd, in_claim):
        r0, _ = self.reduced[nd["idx"]]
        lay, s = nd["layout"], nd["shape"]
        A, B = self.trace[nd["inputs"][0]], self.trace[nd["inputs"][1]]
        if lay == "mk,kn->mn":
            m, k, n = s; b = 1
        elif lay == "k,nk->n":
            k, n = s; b = m = 1
        else:
            b, m, k, n = s
        lb, lm, lk, ln = ilog2(b), ilog2(m), ilog2(k), ilog2(n)
        eq = None; sched = 0; sa = sb = 0
        if lay == "mk,kn->mn":
            r_m, r_n = r0[:lm], r0[lm:]
            eq_m = orc.eq_evals(np.ascontiguousarray(r_m)) if lm else orc.from_ints([1]); eq_n = orc.eq_evals(np.ascontiguousarray(r_n)) if ln else orc.from_ints([1])
            left, right = orc.fr_array(k), orc.fr_array(k)
            orc.lib.orc_fold_i32_cols(A.ctypes.data_as(C.c_void_p), C.c_size_t(m), C.c_size_t(k), orc._p(eq_m), orc._p(left))
            orc.lib.orc_fold_i32_rows(B.ctypes.data_as(C.c_void_p), C.c_size_t(k), C.c_size_t(n), orc._p(eq_n), orc._p(right))
        elif lay == "k,nk->n":
            r_n = r0
            eq_n = orc.eq_evals(np.ascontiguousarray(r_n))
            left = fr(A)
            lo, right = orc.fr_array(k), orc.fr_array(k)
            orc.lib.orc_einsum_fold_layout(C.c_int(4), B.ctypes.data_as(C.c_void_p), B.ctypes.data_as(C.c_void_p), *(C.c_size_t(x) for x in (1, 1, k, n)),
                                           orc._p(eq_n), orc._p(eq_n), orc._p(lo), orc._p(right))
        else:
            code = {"bmk,bkn->mbn": 0, "bmk,kbn->mbn": 1, "mbk,bnk->bmn": 2, "mbk,nbk->bmn": 3}[lay]
            if code <= 1:
                r_m, r_b, r_n = r0[:lm], r0[lm:lm + lb], r0[lm + lb:]; sched, sa, sb = 2, lk, lb
            else:
                r_b, r_m, r_n = r0[:lb], r0[lb:lb + lm], r0[lb + lm:]; sched, sa, sb = 1, lb, lk
            eq_m = orc.eq_evals(np.ascontiguousarray(r_m)) if lm else orc.from_ints([1]); eq_n = orc.eq_evals(np.ascontiguousarray(r_n)) if ln else orc.from_ints([1])
            left, right = orc.fr_array(k * b), orc.fr_array(k * b)
            orc.lib.orc_einsum_fold_layout(C.c_int(code), A.ctypes.data_as(C.c_void_p), B.ctypes.data_as(C.c_void_p), *(C.c_size_t(x) for x in (b, m, k, n)),
                                           orc._p(eq_m), orc._p(eq_n), orc._p(left), orc._p(right))
            eq = orc.eq_evals(np.ascontiguousarray(r_b)) if lb else orc.from_ints([1])
        assert np.array_equal(orc.dot_claim(left, right, eq, sched, sa, sb)[0], in_claim), "einsum input claim"
        proof, ch, fin = orc.sumcheck_dot_prove(left, right, np.ascontiguousarray(in_claim).reshape(1, 4), self.t.t, eq, sched, sa, sb)
        self.proofs[(nd["idx"], PT["EinsumMatmul"])] = [row for row in proof]
        c = orc.challenges_to_fr(ch)
        if lay == "mk,kn->mn":
            lp, rp = np.concatenate([r_m, c]), np.concatenate([c, r_n])
        elif lay == "k,nk->n":
            lp, rp = c, np.concatenate([r_n, c])
        elif lay.startswith("bmk"):
            rj, rh = c[:lk], c[lk:]
            lp = np.concatenate([rh, r_m, rj])
            rp = np.concatenate([rh, rj, r_n]) if lay == "bmk,bkn->mbn" else np.concatenate([rj, rh, r_n])
        else:
            rh, rj = c[:lb], c[lb:]
            lp = np.concatenate([r_m, c])
            rp = np.concatenate([rh, r_n, rj]) if lay == "mbk,bnk->bmn" else np.concatenate([r_n, c])
        self.append_nodeio(nd, 0, lp, fin[0])
        self.append_nodeio(nd, 1, rp, fin[1])

    def op_fused(self, nd):
        i = nd["idx"]
        r0, out_claim = self.reduced[i]
        w = self.wit[i]; S = w["S"]
        eval_R, acc_claim = orc.evaluate(fr(w["rem"]), r0), orc.evaluate(fr(w["quot"]), r0)
        self.append_virtual(node_exec(virt("RescaleRemainder", i), i), r0, eval_R)                     # cache_remainder_prove
        self.append_virtual(node_exec(virt("ClampAcc", i), i), r0, acc_claim)                          # append_raf_claims_prover
        scalar = len(w["rem"]) == 1                                                                    # is_scalar: prove_append_acc only (fused_rebase.rs:215-231)
        cidx = w["quot"].astype(np.int64).view(np.uint64).copy()
        if not scalar:
            self.clamp_lookup(nd, cidx, r0, acc_claim, out_claim)
        in_claim = orc.fr_add_arr(orc.fr_mul_arr(acc_claim, fr([1 << S])[0]), eval_R)                 # fused_input_claim
        if nd["op"] == "Einsum":
            self.einsum_matmul(nd, in_claim)
        else:
            ew = {"Mul": OR.EW_MUL, "Square": OR.EW_SQUARE, "Cube": OR.EW_CUBE}[nd["op"]]
            self.ew_sumcheck(nd, ew, 2 if nd["op"] == "Mul" else 1, in_claim, "RescaleArith")
        if scalar:                                                                                     # ops/mod.rs:604: no remainder range check
            return
        ridx = w["rem"].astype(np.uint64)                                                              # prove_remainder_rc
        phases = 1 if S <= 2 else S // 4 if S % 4 == 0 else S // 2 if S % 2 == 0 else S
        rr_point, rr_claim = self.read_raf(nd, OR.ps_identity(ridx, S, phases, r0), eval_R, ridx, S, "RescaleRemainderRa", "RangeCheck")
        self.onehot_checks(nd, ridx, S, r0, rr_point, rr_claim, "RescaleRemainderRaD", "RescaleRemainderRaChecks")

    def op_addsub(self, nd):
        i = nd["idx"]
        r0, out_claim = self.reduced[i]
        acc = self.wit[i]["acc"]
        if len(acc) > 1:
            acc_claim = orc.evaluate(fr(acc), r0)
            self.append_virtual(node_exec(virt("ClampAcc", i), i), r0, acc_claim)
            self.clamp_lookup(nd, acc.astype(np.int64).view(np.uint64).copy(), r0, acc_claim, out_claim)
        for q in range(2):
            self.append_nodeio(nd, q, r0, orc.evaluate(self.mle(nd["inputs"][q]), r0))

    def op_relu(self, nd):
        i = nd["idx"]
        r0, out_claim = self.reduced[i]
        x = self.trace[nd["inputs"][0]]
        operand_claim = orc.evaluate(fr(x), r0)
        self.append_nodeio(nd, 0, r0, operand_claim)
        gamma = self.t.challenge_scalar()
        lookups = x.astype(np.uint32).astype(np.uint64)
        exec_claim = orc.fr_add_arr(out_claim, orc.fr_mul_arr(gamma, operand_claim))
        if nd["op"] == "Clamp":                                              # ops/clamp.rs: ClampTable<32> = ClampBoundedTable<32, CLAMP_BOUND, true>
            ra_point, ra_claim = self.read_raf(nd, OR.ps_clamp(lookups, 32, CLAMP_BOUND, True, r0, gamma), exec_claim, lookups, 32, "SymmetricClampRa", "Execution")
            self.onehot_checks(nd, lookups, 32, r0, ra_point, ra_claim, "SymmetricClampRaD", "RaOneHotChecks")
            return
        ra_point, ra_claim = self.read_raf(nd, OR.ps_relu(lookups, 32, r0, gamma), exec_claim, lookups, 32, "NodeOutputRa", "Execution")
        self.onehot_checks(nd, lookups, 32, r0, ra_point, ra_claim, "NodeOutputRaD", "RaOneHotChecks")

    def range_check(self, lookups, r_cycle, left, right):
        """ps_read_raf_prover (binary): gamma, UnsignedLessThan over interleave(remainder, bound); claim 1 + gamma left + gamma^2 right"""
        gamma = self.t.challenge_scalar()
        claim = orc.fr_add_arr(one(), orc.fr_add_arr(orc.fr_mul_arr(gamma, left), orc.fr_mul_arr(orc.fr_mul_arr(gamma, gamma), right)))
        return OR.ps_ult(lookups, r_cycle, gamma), claim

    def range_and_onehot(self, nd, lookups, r_cycle, left, right, ra_vp, cp_name, pt_onehot):
        inst, claim = self.range_check(lookups, r_cycle, left, right)
        ra_point, ra_claim = self.read_raf(nd, inst, claim, lookups, 64, ra_vp, "RangeCheck")
        self.onehot_checks(nd, lookups, 64, r_cycle, ra_point, ra_claim, cp_name, pt_onehot)

    def op_sum(self, nd):
        i = nd["idx"]
        r0, out_claim = self.reduced[i]
        acc = self.wit[i]["acc"]
        acc_claim = orc.evaluate(fr(acc), r0)
        self.append_virtual(node_exec(virt("ClampAcc", i), i), r0, acc_claim)
        if len(acc) > 1:
            self.clamp_lookup(nd, acc.astype(np.int64).view(np.uint64).copy(), r0, acc_claim, out_claim)
        idims = list(self.nodes[nd["inputs"][0]]["dims"]); axis = nd["axes"][0]
        while len(idims) > 2:
            assert idims[0] == 1; idims.pop(0); axis -= 1
        if len(idims) == 1:
            idims = [idims[0], 1]
        m, n = idims
        X = fr(self.trace[nd["inputs"][0]]).reshape(m, n, 4)
        eq = orc.eq_evals(np.ascontiguousarray(r0)) if len(r0) else orc.from_ints([1])
        if axis == 0:
            v = np.stack([sum_fr([orc.fr_mul_arr(X[h, j], eq[j]) for j in range(n)]) for h in range(m)])
        else:
            v = np.stack([sum_fr([orc.fr_mul_arr(X[h, j], eq[h]) for h in range(m)]) for j in range(n)])
        I = OR.softmax(OR.SM_SUM_AXIS, v, None, 0, ilog2(len(v)), None)
        rs = self.run(I, acc_claim, i, "SumReduction")
        pt = np.concatenate([rs, r0]) if axis == 0 else np.concatenate([r0, rs])
        self.append_nodeio(nd, 0, pt, I.finals()[0])

    def op_scalar_const_div(self, nd):
        i = nd["idx"]
        r0, claim = self.reduced[i]
        I = OR.elementwise(OR.EW_SUB, [self.mle(nd["inputs"][0]), fr(self.wit[i]["rem"])], r0)
        rs = self.run(I, orc.fr_mul_arr(claim, fr([nd["divisor"]])[0]), i, "Execution")
        fin = I.finals(); pt = np.ascontiguousarray(rs[::-1])
        self.append_nodeio(nd, 0, pt, fin[0])
        self.append_dense(nd, "ScalarConstDivNodeRemainder", pt, fin[1])

    def op_slice(self, nd):
        i = nd["idx"]
        r0, claim = self.reduced[i]
        idims = self.nodes[nd["inputs"][0]]["dims"]
        eq = orc.eq_evals(np.ascontiguousarray(r0))
        sel = orc.fr_array(int(np.prod(idims))).reshape(*idims, 4)
        sl = [slice(None)] * len(idims); sl[nd["axis"]] = slice(nd["start"], nd["end"])
        sel[tuple(sl)] = eq.reshape(*nd["dims"], 4)
        I = OR.elementwise(OR.EW_DOT, [self.mle(nd["inputs"][0]), np.ascontiguousarray(sel.reshape(-1, 4))], orc.fr_array(ilog2(int(np.prod(idims)))))
        rs = self.run(I, claim, i, "Execution")
        self.append_nodeio(nd, 0, np.ascontiguousarray(rs[::-1]), I.finals()[0])

    def op_concat(self, nd):
        """Concat (ops/concat.rs:44-66, 210-372): sum over the inputs of input_t(x) selector_t(x) over the LARGEST input's hypercube, LowToHigh;
        a shorter input is repeated over the low variables (extend_input_to_max_domain), its selector sits at index << shift (build_concat_selector)"""
        i = nd["idx"]
        r0, claim = self.reduced[i]
        eq = orc.eq_evals(np.ascontiguousarray(r0)).reshape(*nd["dims"], 4)
        idims = [self.nodes[j]["dims"] for j in nd["inputs"]]
        nvs = [ilog2(int(np.prod(d))) for d in idims]
        mx = max(nvs)
        ops, off = [], 0
        for k, j in enumerate(nd["inputs"]):
            shift = mx - nvs[k]
            ext = np.repeat(self.mle(j), 1 << shift, axis=0)
            sl = [slice(None)] * len(idims[k]); sl[nd["axis"]] = slice(off, off + idims[k][nd["axis"]])
            off += idims[k][nd["axis"]]
            sel = orc.fr_array(1 << mx)
            sel[:: 1 << shift] = eq[tuple(sl)].reshape(-1, 4)
            ops += [np.ascontiguousarray(ext), sel]
        I = OR.elementwise(OR.EW_DOT, ops, orc.fr_array(mx))
        rs = self.run(I, claim, i, "Execution")
        pt = np.ascontiguousarray(rs[::-1]); fin = I.finals()
        for k in range(len(nd["inputs"])):
            self.append_nodeio(nd, k, np.ascontiguousarray(pt[:nvs[k]]), fin[2 * k])

    def op_div(self, nd):
        i = nd["idx"]
        n = ilog2(len(self.trace[i]))
        r = self.t.challenge_vector_opt(n)
        rem = self.wit[i]["rem"]
        I = OR.elementwise(OR.EW_DIV, [self.mle(nd["inputs"][0]), self.mle(nd["inputs"][1]), self.mle(i), fr(rem)], r)
        rs = self.run(I, orc.fr_array(1)[0], i, "Execution")
        fin = I.finals(); pt = np.ascontiguousarray(rs[::-1])
        self.append_nodeio(nd, 0, pt, fin[0]); self.append_nodeio(nd, 1, pt, fin[1])
        self.append_virtual(node_exec(virt("NodeOutput", i), i), pt, fin[2])
        self.append_advice(nd, "DivRemainder", pt, fin[3])
        self.eval_reduction(nd)
        r0, claim = self.reduced[i]
        self.append_dense(nd, "DivNodeQuotient", r0, claim)
        if len(rem) > 1:
            self.range_and_onehot(nd, interleave_arr(rem, self.trace[nd["inputs"][1]]), pt, fin[3], fin[1], "DivRangeCheckRa", "DivRangeCheckRaD", "RaOneHotChecks")

    def op_mean_of_squares(self, nd):
        i = nd["idx"]
        r0, out_claim = self.reduced[i]
        w = self.wit[i]; D = w["D"]
        eval_R, acc_claim = orc.evaluate(fr(w["rem"]), r0), orc.evaluate(fr(w["quot"]), r0)
        self.append_advice(nd, "RescaleRemainder", r0, eval_R)
        self.append_advice(nd, "ClampAcc", r0, acc_claim)
        if len(w["rem"]) > 1:
            self.clamp_lookup(nd, w["quot"].astype(np.int64).view(np.uint64).copy(), r0, acc_claim, out_claim)
        X = self.mle(nd["inputs"][0])
        log_ret, log_red = len(r0), ilog2(len(X)) - len(r0)
        eq = orc.eq_evals(np.ascontiguousarray(r0)) if log_ret else orc.from_ints([1])
        in_claim = orc.fr_add_arr(orc.fr_mul_arr(acc_claim, fr([D])[0]), eval_R)
        assert np.array_equal(orc.dot_claim(X, X, eq, 1, log_ret, log_red)[0], in_claim), "mean-of-squares input claim"
        proof, ch, fin = orc.sumcheck_dot_prove(X, X.copy(), np.ascontiguousarray(in_claim).reshape(1, 4), self.t.t, eq, 1, log_ret, log_red)
        self.proofs[(i, PT["RescaleArith"])] = [row for row in proof]
        self.append_nodeio(nd, 0, orc.challenges_to_fr(ch), fin[0])
        if len(w["rem"]) > 1:
            lookups = interleave_arr(w["rem"], np.full(len(w["rem"]), D, dtype=np.int32))
            self.range_and_onehot(nd, lookups, r0, eval_R, fr([D])[0], "MeanOfSquaresRangeCheckRa", "MeanOfSquaresRangeCheckRaD", "RescaleRemainderRaChecks")

    def op_rsqrt(self, nd):
        i = nd["idx"]
        w = self.wit[i]
        n = ilog2(len(self.trace[i]))
        r = self.t.challenge_vector_opt(n)
        gamma = self.t.challenge_scalar()
        x = self.trace[nd["inputs"][0]]
        I = OR.elementwise(OR.EW_RSQRT, [fr(x), fr(w["quot"]), self.mle(i), fr(w["div_rem"]), fr(w["sqrt_rem"])], r,
                           constants=np.stack([fr([1 << (3 * nd["scale"])])[0], gamma]))
        rs = self.run(I, orc.fr_array(1)[0], i, "Execution")
        fin = I.finals(); pt = np.ascontiguousarray(rs[::-1])
        self.append_nodeio(nd, 0, pt, fin[0])
        self.append_dense(nd, "RsqrtQuotient", pt, fin[1])
        self.append_virtual(node_exec(virt("NodeOutput", i), i), pt, fin[2])
        self.append_advice(nd, "DivRemainder", pt, fin[3])
        self.append_advice(nd, "SqrtRemainder", pt, fin[4])
        self.eval_reduction(nd)
        lk1, lk2 = interleave_arr(w["div_rem"], x), interleave_arr(w["sqrt_rem"], w["bound"])
        i1, c1 = self.range_check(lk1, pt, fin[3], fin[0])
        i2, c2 = self.range_check(lk2, pt, fin[4], orc.fr_add_arr(orc.fr_add_arr(fin[2], fin[2]), one()))
        rows, ch, _ = OB.batched_prove([OB.ra_instance(i1, c1), OB.ra_instance(i2, c2)], self.t.t)
        self.proofs[(i, PT["RangeCheck"])] = rows
        rs2 = orc.challenges_to_fr(ch)
        ra_point = np.concatenate([rs2[:64], rs2[64:][::-1]])
        fams = []
        for lk, vp, cp in ((lk1, "DivRangeCheckRa", "SqrtDivRangeCheckRaD"), (lk2, "SqrtRangeCheckRa", "SqrtRangeCheckRaD")):
            ra_claim = OR.ra_claim(lk, 64, ra_point)
            self.append_virtual(node_exec(virt(vp, i), i), ra_point, ra_claim)
            fams.append((lk, 64, pt, ra_point, ra_claim, cp))
        self.onehot_checks_multi(nd, fams, "RaOneHotChecks")

    def op_trig(self, nd):
        """Sin / Cos by neural teleportation (ops/sin.rs:56-186; ReductionFlow::Custom): division by the period modulus at a fresh point, the
        downscale right-shift lookup batched with the table read-raf, the downscale one-hot checks, the committed quotient, the eval reduction,
        prove_range_and_onehot (neural_teleport/range_and_onehot.rs:68-140)"""
        i = nd["idx"]; w = self.wit[i]; op = nd["op"]
        n = ilog2(len(self.trace[i])); LK = TRIG_TABLE_VARS; K = 1 << LK
        tau = fr([TRIG_PERIOD_MODULUS])[0]
        r = self.t.challenge_vector_opt(n)                                                                  # TeleportDivisionParams::new_from_transcript
        x = self.trace[nd["inputs"][0]]
        I = OR.elementwise(OR.EW_TELEPORT_DIV, [fr(x), fr(w["quot"]), fr(w["rem"])], r, constants=tau.reshape(1, 4))
        rs = self.run(I, orc.fr_array(1)[0], i, "NeuralTeleport")
        fin = I.finals(); pt = np.ascontiguousarray(rs[::-1])
        self.append_nodeio(nd, 0, pt, fin[0])
        self.append_advice(nd, "TeleportQuotient", pt, fin[1])
        self.append_advice(nd, "TeleportRemainder", pt, fin[2])
        q_claim, rem_claim = fin[1], fin[2]
        down_claim = orc.evaluate(fr(w["down"]), pt)
        self.append_advice(nd, "TrigDownscaled", pt, down_claim)                                            # cache_downscaled_prove
        self.append_advice(nd, "TeleportRemainder", pt, rem_claim)                                          # append_raf_claims_prover of the downscale lookup
        g_d = self.t.challenge_scalar()
        lk_rem = w["rem"].astype(np.uint32).astype(np.uint64)
        I_dsc = OR.ps_rshift(lk_rem, 32, TRIG_DOWNSCALE_BITS, pt, g_d)
        g_s = self.t.challenge_scalar()                                                                     # SinParams::new
        out_claim = orc.evaluate(self.mle(i), pt)
        self.append_virtual(node_exec(virt("NodeOutput", i), i), pt, out_claim)                             # SinProver::initialize: Target::Current
        lk_down = w["down"].astype(np.uint64)
        ra = self.ra_histogram(lk_down, K, pt)
        I_tab = OR.elementwise(OR.EW_GATHER, [ra, fr_fast(trig_table(op)), fr_fast(np.arange(K, dtype=np.int64))], orc.fr_array(LK), constants=g_s.reshape(1, 4))
        c_dsc = orc.fr_add_arr(down_claim, orc.fr_mul_arr(g_d, rem_claim)); c_tab = orc.fr_add_arr(out_claim, orc.fr_mul_arr(g_s, down_claim))
        rows, ch, _ = OB.batched_prove([OB.ra_instance(I_dsc, c_dsc), OB.ra_instance(I_tab, c_tab)], self.t.t)
        self.proofs[(i, PT["Execution"])] = rows
        rs2 = orc.challenges_to_fr(ch); mr = len(rs2)
        Dra_point, Dra_claim = self.ra_opening(nd, "TrigDownscaleRa", lk_rem, 32, rs2)
        tab_point = np.concatenate([rs2[mr - LK:][::-1], pt]); tab_claim = I_tab.finals()[0]
        self.append_advice(nd, "SinRa" if op == "Sin" else "CosRa", tab_point, tab_claim)
        self.onehot_checks(nd, lk_rem, 32, pt, Dra_point, Dra_claim, "TrigDownscaleRaD", "TrigDownscaleRaChecks")
        self.append_dense(nd, "TeleportNodeQuotient", pt, q_claim)
        self.eval_reduction(nd)
        lk_rc = interleave_arr(w["rem"], np.full(len(w["rem"]), TRIG_PERIOD_MODULUS, dtype=np.int32))
        I_rc, c_rc = self.range_check(lk_rc, pt, rem_claim, tau)
        oh, st = self.onehot_build([(lk_down, LK, pt, tab_point, tab_claim, "SinRaD" if op == "Sin" else "CosRaD")])
        rows, ch, _ = OB.batched_prove([OB.ra_instance(I_rc, c_rc)] + oh, self.t.t)
        self.proofs[(i, PT["RaOneHotChecks"])] = rows
        rs3 = orc.challenges_to_fr(ch)
        Rra_point, Rra_claim = self.ra_opening(nd, "TeleportRangeCheckRa", lk_rc, 64, rs3)
        self.onehot_cache(i, st, rs3)
        self.onehot_checks(nd, lk_rc, 64, pt, Rra_point, Rra_claim, "TeleportRangeCheckRaD", "RaHammingWeight")

    def ra_histogram(self, lookups, K, r):
        """compute_ra_evals: ra[k] = sum_{j : idx_j = k} eq(r, j)"""
        E = orc.eq_evals(np.ascontiguousarray(r)) if len(r) else orc.from_ints([1])
        out = orc.fr_array(K)
        for j, k in enumerate(lookups):
            out[int(k)] = orc.fr_add_arr(out[int(k)], E[j])
        return out

    def op_tanh(self, nd):
        i = nd["idx"]
        r0, out_claim = self.reduced[i]
        w = self.wit[i]
        LK = ACTIVATION_TABLE_VARS; K = 1 << LK
        gamma = self.t.challenge_scalar()
        clamped_claim = orc.evaluate(fr(w["clamped"]), r0)
        self.append_advice(nd, "ActivationClampedOutput", r0, clamped_claim)
        ra = self.ra_histogram(w["small_idx"], K, r0)
        ident = np.arange(K, dtype=np.int64); ident[K // 2:] -= K
        I = OR.elementwise(OR.EW_GATHER, [ra, fr_fast(activation_table(nd["op"])), fr_fast(ident)], orc.fr_array(LK), constants=gamma.reshape(1, 4))
        rs = self.run(I, orc.fr_add_arr(out_claim, orc.fr_mul_arr(gamma, clamped_claim)), i, "Execution")
        small_pt = np.concatenate([rs[::-1], r0])
        ra_small = I.finals()[0]
        self.append_advice(nd, "ActivationSmallRa", small_pt, ra_small)
        x = self.trace[nd["inputs"][0]]
        operand_claim = orc.evaluate(fr(x), r0)
        self.append_nodeio(nd, 0, r0, operand_claim)
        gamma2 = self.t.challenge_scalar()
        lookups = x.astype(np.uint32).astype(np.uint64)
        claim2 = orc.fr_add_arr(clamped_claim, orc.fr_mul_arr(gamma2, operand_claim))
        ra_point, ra_claim = self.read_raf(nd, OR.ps_clamp(lookups, 32, ACTIVATION_BOUND, True, r0, gamma2), claim2, lookups, 32, "ActivationClampRa", "NeuralTeleport")
        self.onehot_checks_multi(nd, [(w["small_idx"], LK, r0, small_pt, ra_small, "ActivationSmallRaD"),
                                      (lookups, 32, r0, ra_point, ra_claim, "ActivationClampRaD")], "RaOneHotChecks")

    def op_gather(self, nd):
        i = nd["idx"]
        r0, out_claim = self.reduced[i]
        ddims = self.nodes[nd["inputs"][0]]["dims"]
        V = ddims[0]; word = int(np.prod(ddims)) // V
        idx = self.trace[nd["inputs"][1]]
        ln, lv = ilog2(len(idx)), ilog2(V)
        gamma = self.t.challenge_scalar()
        r_index, r_word = r0[:ln], r0[ln:]
        index_claim = orc.evaluate(fr(idx), np.ascontiguousarray(r_index))
        self.append_nodeio(nd, 1, r_index, index_claim)
        lookups = idx.astype(np.uint32).astype(np.uint64)
        ra = self.ra_histogram(lookups, V, r_index)
        eq_w = orc.eq_evals(np.ascontiguousarray(r_word)) if len(r_word) else orc.from_ints([1])
        D = fr(self.trace[nd["inputs"][0]]).reshape(V, word, 4)
        dict_r = np.stack([sum_fr([orc.fr_mul_arr(D[k, wv], eq_w[wv]) for wv in range(word)]) for k in range(V)])
        I = OR.elementwise(OR.EW_GATHER, [ra, dict_r, fr(np.arange(V))], orc.fr_array(lv), constants=gamma.reshape(1, 4))
        rs = self.run(I, orc.fr_add_arr(out_claim, orc.fr_mul_arr(gamma, index_claim)), i, "Execution")
        fin = I.finals()
        pt = rs[::-1]
        ra_pt, dict_pt = np.concatenate([pt, r_index]), np.concatenate([pt, r_word])
        self.append_advice(nd, "NodeOutputRa", ra_pt, fin[0])
        self.append_nodeio(nd, 0, dict_pt, fin[1])
        if nd["op"] == "GatherSmall":
            return self.gather_small_checks(nd, idx.astype(np.int32), np.ascontiguousarray(r_index), lv, ln)
        self.onehot_checks(nd, lookups, lv, np.ascontiguousarray(r_index), ra_pt, fin[0], "GatherRaD", "RaOneHotChecks")

    def gather_small_checks(self, nd, rows, r_index, lv, ln):
        """GatherSmall after the execution sumcheck (ops/gather/small.rs:44-62, 124-168, 216-255, 290-312, 358-377): BatchedSumcheck
        [HammingBooleanity of hw = [1; N] (d = 1, gamma_powers [1], r_cycle = the index operand's point), Booleanity with d = 1,
        log_k_chunk = log dict_len, gammas = [Challenge::from(1)]], then HammingWeight (d = 1, gamma_powers [1]) on its own"""
        i = nd["idx"]
        r_address = self.t.challenge_vector_opt(lv)                          # ra_booleanity_params
        G = OR.ra_G([rows], lv, r_index)                                     # compute_ra_evals(r_cycle, indexes, num_words)
        hw = np.stack([one()] * (1 << ln))
        gamma_b = orc.challenges_to_fr([1])                                  # F::Challenge::from(1)
        I_hb = OR.elementwise(OR.EW_HAMMING_BOOL, [hw], r_index, constants=one().reshape(1, 4))
        I_bo = OR.booleanity(G, [rows], lv, gamma_b, r_address, r_index)
        zero = orc.fr_array(1)[0]
        proof, ch, _ = OB.batched_prove([OB.ra_instance(I_hb, zero), OB.ra_instance(I_bo, zero)], self.t.t)
        self.proofs[(i, PT["RaOneHotChecks"])] = proof
        rs = orc.challenges_to_fr(ch); mr = len(rs)
        p_hb = np.ascontiguousarray(rs[mr - ln:][::-1])
        self.append_virtual(oid(virt("HammingWeight"), "RamHammingBooleanity"), p_hb, I_hb.finals()[0])
        ba = np.ascontiguousarray(rs[:lv][::-1]); bc = np.ascontiguousarray(rs[lv:][::-1])
        Fb = orc.eq_evals(ba)
        self.append_sparse("GatherRa", i, 0, "Booleanity", np.concatenate([ba, bc]), orc.evaluate(np.ascontiguousarray(Fb[rows]), bc))
        I_hw = OR.hamming(G, lv, one().reshape(1, 4))
        rs3 = self.run(I_hw, one(), i, "RaHammingWeight")
        hw_rs = np.ascontiguousarray(rs3[::-1])
        self.append_sparse("GatherRa", i, 0, "HammingWeight", np.concatenate([hw_rs, r_index]), orc.evaluate(G[0], hw_rs))

    def ra_opening(self, nd, vp, lookups, log_K, sl):
        """cache_openings of a PS-Shout / IdentityRC instance whose challenge slice is `sl`: ra at (address challenges, reversed cycle challenges)"""
        ra_point = np.concatenate([sl[:log_K], sl[log_K:][::-1]])
        ra_claim = OR.ra_claim(lookups, log_K, ra_point)
        self.append_advice(nd, vp, ra_point, ra_claim)
        return ra_point, ra_claim

    def op_softmax(self, nd):
        """SoftmaxLastAxisProver::prove (ops/softmax_last_axis/mod.rs:177-262): auxiliary vectors, then four BatchedSumcheck stages"""
        i = nd["idx"]
        r0, out_claim = self.reduced[i]
        w = self.wit[i]
        F, N, S = w["F"], w["N"], w["S"]
        lf, ln, LS = ilog2(F), ilog2(N), MODEL_SCALE
        # ONE row ([1, N]) and ONE element ([1, 1]) run the generic flow.  SEVERAL rows of ONE element ([F, 1], F >= 2) are not provable by the
        # reference either: ExpSumProver / MaxIndicatorProver create their split eq at `round == log_N() - 1` (exp_sum.rs:187, max.rs:251) — with
        # log_N = 0 the usize subtraction overflows (a panic in debug; in release it wraps, the Option stays None and compute_phase_2_message's
        # unwrap() panics at exp_sum.rs:163 / max.rs:217) while their log_F phase-2 rounds do run.  Refused here as the reference refuses it.
        if ln == 0 and lf > 0:
            raise ValueError("SoftmaxLastAxis over several rows of ONE element: the reference's ExpSum / MaxIndicator provers panic (log_N() - 1 at log_N = 0)")
        log_T = lf + ln
        u = lambda a: a.astype(np.uint32).astype(np.uint64)
        phases = LS // 4 if LS % 4 == 0 else LS // 2
        empty = orc.fr_array(0)
        for k in range(F):                                                   # send_auxiliary_vectors (:392-413): F::from_u32(v as u32)
            self.append_virtual(node_exec(virt("SoftmaxSumOutput", i, k), i), empty, fr([int(np.uint32(w["exp_sum_q"][k]))])[0])
            self.append_virtual(node_exec(virt("SoftmaxMaxOutput", i, k), i), empty, fr([int(np.uint32(w["max_k"][k]))])[0])
            self.append_virtual(node_exec(virt("SoftmaxMaxIndex", i, k), i), empty, fr([int(np.uint32(w["argmax_k"][k]))])[0])
        r_lead = np.ascontiguousarray(r0[:lf])
        exp_sum_claim = orc.evaluate(fr(w["exp_sum_q"]), r_lead)
        self.append_advice(nd, "SoftmaxExpSum", r_lead, exp_sum_claim)                                       # cache_exp_sum
        R_claim = orc.evaluate(fr(w["R"]), r0)
        self.append_advice(nd, "SoftmaxRecipMultRemainder", r0, R_claim)                                     # cache_R
        # ---- stage 1: RecipMult, ExpSum, IdentityRC(R)
        exp_q = fr(w["exp_q"])
        insts = [OB.ra_instance(OR.softmax(OR.SM_RECIP_MULT, exp_q, fr(w["inv_sum"]), lf, ln, r0), orc.fr_add_arr(orc.fr_mul_arr(out_claim, fr([S])[0]), R_claim)),
                 OB.ra_instance(OR.softmax(OR.SM_EXP_SUM, exp_q, None, lf, ln, r_lead), exp_sum_claim),
                 OB.ra_instance(OR.ps_identity(u(w["R"]), LS, phases, r0), R_claim)]
        rows, ch, _ = OB.batched_prove(insts, self.t.t)
        self.proofs[(i, PT["SoftmaxStage1"])] = rows
        rs = orc.challenges_to_fr(ch); mr = len(rs)
        r1 = np.ascontiguousarray(rs[mr - log_T:][::-1])
        exp_q_claim = orc.evaluate(exp_q, r1)
        self.append_advice(nd, "SoftmaxExpQ", r1, exp_q_claim)                                               # RecipMult::cache_openings
        self.append_advice(nd, "SoftmaxExpQ", r1, exp_q_claim)                                               # ExpSum::cache_openings
        Rra_point, Rra_claim = self.ra_opening(nd, "SoftmaxRemainderRa", u(w["R"]), LS, rs[mr - LS - log_T:])
        # ---- stage 2: Mult, MaxIndicator, IdentityRC(r_exp), one-hot checks of R
        r_exp_claim = orc.evaluate(fr(w["r_exp"]), r1)
        self.append_advice(nd, "SoftmaxExpRemainder", r1, r_exp_claim)                                       # cache_r_exp
        max_k_eval = orc.evaluate(fr(w["max_k"]), np.ascontiguousarray(r1[:lf]))
        e = np.zeros(F * N, dtype=np.int32); e[np.arange(F) * N + w["argmax_k"]] = 1
        X = self.mle(nd["inputs"][0])
        insts = [OB.ra_instance(OR.elementwise(OR.EW_MUL, [fr(w["exp_hi"]), fr(w["exp_lo"])], r1), orc.fr_add_arr(orc.fr_mul_arr(exp_q_claim, fr([S])[0]), r_exp_claim)),
                 OB.ra_instance(OR.softmax(OR.SM_MAX_INDICATOR, X, fr(e), lf, ln, np.ascontiguousarray(r1[:lf])), max_k_eval),
                 OB.ra_instance(OR.ps_identity(u(w["r_exp"]), LS, phases, r1), r_exp_claim)]
        oh, st = self.onehot_build([(u(w["R"]), LS, r0, Rra_point, Rra_claim, "SoftmaxRemainderRaD")])
        rows, ch, _ = OB.batched_prove(insts + oh, self.t.t)
        self.proofs[(i, PT["SoftmaxStage2"])] = rows
        rs = orc.challenges_to_fr(ch); mr = len(rs)
        r2 = np.ascontiguousarray(rs[mr - log_T:][::-1])
        self.append_advice(nd, "SoftmaxExpHi", r2, orc.evaluate(fr(w["exp_hi"]), r2))                        # Mult::cache_openings
        self.append_advice(nd, "SoftmaxExpLo", r2, orc.evaluate(fr(w["exp_lo"]), r2))
        self.append_nodeio(nd, 0, r2, orc.evaluate(X, r2))                                                   # MaxIndicator::cache_openings
        Era_point, Era_claim = self.ra_opening(nd, "SoftmaxExpRemainderRa", u(w["r_exp"]), LS, rs[mr - LS - log_T:])
        self.onehot_cache(i, st, rs)
        # ---- stage 3: the exp-digit Shout lookups, the significance clamp lookup, one-hot checks of r_exp
        exp_hi_claim, exp_lo_claim = self.openings[node_exec(virt("SoftmaxExpHi", i), i)][1], self.openings[node_exec(virt("SoftmaxExpLo", i), i)][1]
        z_hi_claim, z_lo_claim = orc.evaluate(fr(w["z_hi"]), r2), orc.evaluate(fr(w["z_lo"]), r2)
        self.append_advice(nd, "SoftmaxZHi", r2, z_hi_claim)                                                 # cache_z_hi_lo
        self.append_advice(nd, "SoftmaxZLo", r2, z_lo_claim)
        lk_hi, lk_lo = ilog2(len(w["lut_hi"])), ilog2(len(w["lut_lo"]))
        g_hi = self.t.challenge_scalar()                                                                     # ReadRafParams::new (shout.rs:112-130)
        I_hi = OR.shout_read_raf(u(w["z_hi"]), w["lut_hi"], lk_hi, r2, g_hi)
        g_lo = self.t.challenge_scalar()
        I_lo = OR.shout_read_raf(u(w["z_lo"]), w["lut_lo"], lk_lo, r2, g_lo)
        z_claim = orc.evaluate(fr(w["z"]), r2)
        self.append_advice(nd, "SoftmaxClampWitness", r2, z_claim)                                           # append_raf_claims_prover (op_lookups/mod.rs:404-418)
        g_c = self.t.challenge_scalar()                                                                      # ps_read_raf_prover (unary.rs:112)
        rv = orc.fr_add_arr(orc.fr_mul_arr(z_hi_claim, fr([1 << w["log2_base"]])[0]), z_lo_claim)            # significance_clamp.rs:61-69
        bound = lk_hi + w["log2_base"]                                                                        # SOFTMAX_CLAMP_BOUND = log2(padded hi_size * base)
        insts = [OB.ra_instance(I_hi, orc.fr_add_arr(exp_hi_claim, orc.fr_mul_arr(g_hi, z_hi_claim))),
                 OB.ra_instance(I_lo, orc.fr_add_arr(exp_lo_claim, orc.fr_mul_arr(g_lo, z_lo_claim))),
                 OB.ra_instance(OR.ps_clamp(u(w["z"]), 32, bound, False, r2, g_c), orc.fr_add_arr(rv, orc.fr_mul_arr(g_c, z_claim)))]
        oh, st = self.onehot_build([(u(w["r_exp"]), LS, r1, Era_point, Era_claim, "SoftmaxExpRemainderRaD")])
        rows, ch, _ = OB.batched_prove(insts + oh, self.t.t)
        self.proofs[(i, PT["SoftmaxStage3"])] = rows
        rs = orc.challenges_to_fr(ch); mr = len(rs)
        hi_point = np.concatenate([rs[mr - lk_hi:], r2]); hi_claim = I_hi.final()                            # ReadRafProver::cache_openings: [challenges | r]
        self.append_advice(nd, "SoftmaxZHiRa", hi_point, hi_claim)
        lo_point = np.concatenate([rs[mr - lk_lo:], r2]); lo_claim = I_lo.final()
        self.append_advice(nd, "SoftmaxZLoRa", lo_point, lo_claim)
        Cra_point, Cra_claim = self.ra_opening(nd, "SoftmaxClampRa", u(w["z"]), 32, rs[mr - 32 - log_T:])
        self.onehot_cache(i, st, rs)
        # ---- stage 4: one-hot checks of z_hi, z_lo and the clamp lookup
        self.onehot_checks_multi(nd, [(u(w["z_hi"]), lk_hi, r2, hi_point, hi_claim, "SoftmaxZHiRaD"), (u(w["z_lo"]), lk_lo, r2, lo_point, lo_claim, "SoftmaxZLoRaD"),
                                      (u(w["z"]), 32, r2, Cra_point, Cra_claim, "SoftmaxClampRaD")], "SoftmaxStage4")

    def prove_node(self, nd):
        op, i = nd["op"], nd["idx"]
        if op == "Div":
            return self.op_div(nd)                                           # ReductionFlow::Custom
        if op == "Rsqrt":
            return self.op_rsqrt(nd)
        if op in ("Sin", "Cos"):
            return self.op_trig(nd)
        self.eval_reduction(nd)
        r0, claim = self.reduced[i]
        if op in ("Input", "Constant"):
            return
        if op == "Concat":
            return self.op_concat(nd)
        if op in ("Erf", "Sigmoid"):
            return self.op_tanh(nd)
        if op == "GatherSmall":
            return self.op_gather(nd)                                          # prove_clamped_activation::<_, _, Table>
        if op in ("Sum", "ScalarConstDiv", "Slice", "MeanOfSquares", "Tanh", "GatherLarge", "SoftmaxLastAxis"):
            return {"Sum": self.op_sum, "ScalarConstDiv": self.op_scalar_const_div, "Slice": self.op_slice, "MeanOfSquares": self.op_mean_of_squares,
                    "Tanh": self.op_tanh, "GatherLarge": self.op_gather, "SoftmaxLastAxis": self.op_softmax}[op](nd)
        if op == "Identity":
            self.append_nodeio(nd, 0, r0, claim)
        elif op in ("Add", "Sub"):
            self.op_addsub(nd)
        elif op in ("Einsum", "Mul", "Square", "Cube"):
            self.op_fused(nd)
        elif op == "And":
            self.ew_sumcheck(nd, OR.EW_MUL, 2, claim, "Execution")
        elif op == "Iff":
            self.ew_sumcheck(nd, OR.EW_IFF, 3, claim, "Execution")
        elif op in ("ReLU", "Clamp"):
            self.op_relu(nd)
        elif op in ("Neg", "IsNan"):                                         # ops/neg.rs, is_nan.rs: no sumcheck, the operand at the reduced point
            self.append_nodeio(nd, 0, r0, orc.evaluate(self.mle(nd["inputs"][0]), r0))
        elif op == "Reshape":
            I = OR.elementwise(OR.EW_DOT, [self.mle(nd["inputs"][0]), orc.eq_evals(np.ascontiguousarray(r0))], r0)
            rs = self.run(I, claim, i, "Execution")
            self.append_nodeio(nd, 0, np.ascontiguousarray(rs[::-1]), I.finals()[0])
        elif op == "MoveAxis":
            groups, off = [], 0
            for dim in nd["dims"]:
                v = ilog2(dim); groups.append(r0[off:off + v]); off += v
            g = groups.pop(nd["destination"])
            groups.insert(nd["source"], g)
            self.append_nodeio(nd, 0, np.concatenate(groups) if groups else r0, claim)
        elif op == "Broadcast":
            idims = self.nodes[nd["inputs"][0]]["dims"]
            offd = len(nd["dims"]) - len(idims)
            parts, pos = [], 0
            for a, dim in enumerate(nd["dims"]):
                v = ilog2(dim)
                if a >= offd and idims[a - offd] == dim:
                    parts.append(r0[pos:pos + v])
                pos += v
            r_in = np.concatenate(parts) if parts else orc.fr_array(0)
            self.append_nodeio(nd, 0, r_in, orc.evaluate(self.mle(nd["inputs"][0]), np.ascontiguousarray(r_in)))
        else:
            raise ValueError(op)

    def reduced_openings(self):
        if not self.committed:
            self.ro = None
            return
        keys = sorted(self.committed)
        insts = []
        for k in keys:
            c = self.committed[k]
            assert "point" in c, f"committed polynomial {k} never opened"
            if "dense" in c:
                insts.append(OB.ra_instance(OR.dense_opening(c["dense"].copy(), c["point"]), c["claim"]))
            else:
                lk = c.get("lk", 4)
                insts.append(OB.ra_instance(OR.onehot_opening(c["row"], lk, c["point"][:lk], c["point"][lk:]), c["claim"]))
        rows, ch, _ = OB.batched_prove(insts, self.t.t)
        rs = orc.challenges_to_fr(ch)
        fin = []
        for k in keys:
            c = self.committed[k]
            if "dense" in c:
                fin.append(orc.evaluate(c["dense"], np.ascontiguousarray(rs[len(rs) - c["log_T"]:])))
                continue
            lk = c.get("lk", 4)
            sl = rs[len(rs) - lk - c["log_T"]:]
            Fs = orc.eq_evals(np.ascontiguousarray(sl[:lk]))
            fin.append(orc.evaluate(np.stack([Fs[x] for x in c["row"]]), np.ascontiguousarray(sl[lk:])))
        fin = np.stack(fin)
        self.t.append_scalars(fin)
        q = self.t.challenge_scalar()
        gam = [one()]
        for _ in range(1, len(fin)):
            gam.append(orc.fr_mul_arr(gam[-1], q))
        dense = [(self.committed[k]["dense"], g) for k, g in zip(keys, gam) if "dense" in self.committed[k]]
        onehot = [(self.committed[k]["row"], 1 << self.committed[k].get("lk", 4), g) for k, g in zip(keys, gam) if "dense" not in self.committed[k]]
        joint = OB.rlc_build(dense, onehot)
        assert len(joint) == 1 << len(ch)
        com, w, v = orc.hyperkzg_open(self.srs, joint, ch, self.t.t)
        self.ro = dict(rows=rows, claims=fin, com=com, w=w, v=v, ch=ch, joint=joint)

    def prove(self, inputs):
        nodes = [self.nodes[i] for i in self.order]
        self.trace, self.wit = execute(nodes, inputs)
        in_nodes = [nd for nd in nodes if nd["op"] == "Input"]
        t = self.t
        t.append_message(b"model_inputs"); t.append_u64(len(in_nodes)); t.append_u64(len(in_nodes))          # append_inputs_to_transcript
        for nd, x in zip(in_nodes, inputs):
            t.append_u64(nd["idx"]); t.append_u64(len(nd["dims"]))
            for d in nd["dims"]:
                t.append_u64(d)
            t.append_bytes(np.ascontiguousarray(x, dtype="<i4").tobytes())
        self.commit()
        self.output_claim()
        for i in reversed(self.order):
            self.prove_node(self.nodes[i])
        self.reduced_openings()
        return self.serialize()

    def serialize(self):
        """ONNXProof::serialize_compressed (proof_serialization.rs:200-224)"""
        out = u64(len(self.openings))
        for k in sorted(self.openings):
            out += opening_id_bytes(k) + fr_bytes(self.openings[k][1])
        out += u64(len(self.proofs))
        for k in sorted(self.proofs):
            out += u64(k[0]) + bytes([k[1]]) + sumcheck_proof_bytes(self.proofs[k])
        keys = sorted(self.committed)
        out += u64(len(keys)) + b"".join(g1_compressed(self.committed[k]["commitment"]) for k in keys)
        out += u64(len(self.evalred))
        for k in sorted(self.evalred):
            h = self.evalred[k]
            out += u64(k) + u64(len(h)) + b"".join(fr_bytes(c) for c in h)
        if self.ro is None:
            return out + b"\x00"
        ro = self.ro
        out += b"\x01" + sumcheck_proof_bytes(ro["rows"]) + u64(len(ro["claims"])) + b"".join(fr_bytes(c) for c in ro["claims"])
        ell = len(ro["ch"])
        out += u64(len(ro["com"])) + b"".join(g1_compressed(p) for p in ro["com"])                            # HyperKZGProof { com, w, v }
        out += u64(3) + b"".join(g1_compressed(p) for p in ro["w"])
        out += u64(3)
        for i in range(3):
            out += u64(ell) + b"".join(fr_bytes(c) for c in ro["v"][i])
        return out


def interleave_arr(x, y):
    """interleave_bits(x as u32, y as u32) per element (joltworks utils/mod.rs:146-164): x on the odd bit positions, y on the even ones"""
    def spread(v):
        v = np.asarray(v).astype(np.int64).astype(np.uint32).astype(np.uint64)
        v = (v | (v << np.uint64(16))) & np.uint64(0x0000FFFF0000FFFF)
        v = (v | (v << np.uint64(8))) & np.uint64(0x00FF00FF00FF00FF)
        v = (v | (v << np.uint64(4))) & np.uint64(0x0F0F0F0F0F0F0F0F)
        v = (v | (v << np.uint64(2))) & np.uint64(0x3333333333333333)
        v = (v | (v << np.uint64(1))) & np.uint64(0x5555555555555555)
        return v
    return (spread(x) << np.uint64(1)) | spread(y)


def sum_fr(xs):
    acc = orc.fr_array(1)[0]
    for x in xs:
        acc = orc.fr_add_arr(acc, x)
    return acc


def eq_bits(r, value, nbits):
    """prod_i (bit_i ? r_i : 1 - r_i), r[0] <-> MSB of value"""
    w = one()
    minus1 = orc.from_ints([FR - 1])[0]
    for i in range(nbits):
        bit = (int(value) >> (nbits - 1 - i)) & 1
        f = r[i] if bit else orc.fr_add_arr(one(), orc.fr_mul_arr(minus1, r[i]))
        w = orc.fr_mul_arr(w, f)
    return w
