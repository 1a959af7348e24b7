"""Graph descriptions for atlas_prove_graph / oracle.graph (the tracer's ComputationGraph vocabulary,
atlas-onnx-tracer/src/node/mod.rs:12-24, ops/mod.rs:117-155), built with synthetic random-init weights:

  transformer(...)      a GPT-style decoder stack with the operator decomposition the tracer produces for nanoGPT / GPT-2
                        (atlas-onnx-tracer/models/nanoGPT/network.onnx: MatMul -> Einsum, Pow -> Square / Cube, ReduceMean ->
                        Sum + ScalarConstDiv / MeanOfSquares, Sqrt + Div -> Rsqrt, Where -> Iff, tanh-GELU -> Cube / Mul / Tanh,
                        Softmax -> SoftmaxLastAxis, Gather -> GatherSmall for dictionaries of at most 2^16 words, handlers/index.rs:33-45), shapes padded to powers of two (vocab 65 -> 128)
  microgpt()            seq 16, d_model 16, 4 heads, 1 layer, vocab 32; RMSNorm, ReLU MLP, no biases   (BASELINE config 1 shape)
  nanogpt()             seq 64, d_model 64, 4 heads, 4 layers, vocab 128   (BASELINE config 3 shape)
  gpt2_layer()          seq 16, d_model 768 -> 1024, 12 -> 16 heads, one layer + lm-head slice (BASELINE config 4 shape, one layer)
  gpt2()                the same with all 12 layers and the whole (padded) vocabulary: 2^16-word embedding and lm head

`level` selects how much of the decomposition is emitted, following what the graph prover composes:
  0  linear algebra + residuals only (Einsum / Add / Mul / ReLU / Reshape / Iff)      [round-3 first slice]
  1  + LayerNorm (Sum, ScalarConstDiv, Sub, Square, MeanOfSquares / Rsqrt)
  2  + SoftmaxLastAxis, tanh-GELU (Cube, Tanh), the embedding gather (GatherSmall, as the tracer picks it for these vocabularies)

The tensors are synthetic (there is no tracer here); the operator list per layer and the shapes are the model's."""
import numpy as np


class B:
    def __init__(self, seed):
        self.nodes, self.rng = [], np.random.default_rng(seed)

    def add(self, op, inputs, dims, **kw):
        nd = {"idx": len(self.nodes), "op": op, "inputs": list(inputs), "dims": list(dims)}
        nd.update(kw)
        self.nodes.append(nd)
        return nd["idx"]

    def const(self, dims, lo, hi):
        return self.add("Constant", [], dims, data=self.rng.integers(lo, hi, size=int(np.prod(dims))).astype(np.int32))

    def const_data(self, dims, data):
        return self.add("Constant", [], dims, data=np.ascontiguousarray(data, dtype=np.int32).reshape(-1))

    def matmul(self, x, w, m, k, n, S):
        return self.add("Einsum", [x, w], [m, n], layout="mk,kn->mn", scale=S, shape=[m, k, n])


def transformer(layers, seq, d_model, heads, vocab, S=14, level=2, seed=0, mlp_mult=4, final_head=True, norm_kind="layer", act="gelu",
                bias=True, norm0=False, final_norm=True, att_scale=False, mask_fill=None, fused_qkv=False, cube_as_muls=False, weights=None, round_consts=False):
    """weights (optional): the model file's tensors quantized at scale S — {"wte": [vocab][d], "wpe": [seq][d], "ln": [vectors in order of use],
    "mats": [matrices [in][out] in order of use]}, each padded to powers of two with zeros (tests/golden/ref_models.npz, tools/extract_ref_model.py);
    None: random-init.  round_consts: the scalar constants of the graph rounded as quantize_float rounds (utils/quantize.rs:137-185) instead of truncated."""
    b = B(seed)
    wq_ = {k: list(v) if isinstance(v, (list, tuple)) else v for k, v in (weights or {}).items()}
    qc = (lambda f: int(np.floor(abs(f) * (1 << S) + 0.5)) * (1 if f >= 0 else -1)) if (round_consts or weights is not None) else (lambda f: int(f * (1 << S)))

    def mat(k, n):
        if weights is None:
            return b.const([k, n], -wlim, wlim)
        w = np.asarray(wq_["mats"].pop(0), dtype=np.int32)
        assert w.shape == (k, n), (w.shape, k, n)
        return b.const_data([k, n], w)

    def lnw():
        if weights is None:
            return b.const([d_model], one // 2, one + one // 2)
        w = np.asarray(wq_["ln"].pop(0), dtype=np.int32)
        assert w.shape == (d_model,)
        return b.const_data([d_model], w)
    hd = d_model // heads
    wlim = 1 << (S - 2)                      # weights ~ U(-0.25, 0.25) at scale S
    ff = mlp_mult * d_model
    one = 1 << S
    if level >= 2:                           # token + position embedding: Gather(wte, tokens) + wpe
        tok = b.add("Input", [], [seq])
        wte = b.const([vocab, d_model], -one, one) if weights is None else b.const_data([vocab, d_model], np.asarray(wq_["wte"], dtype=np.int32))
        x = b.add("GatherSmall" if vocab <= 65536 else "GatherLarge", [wte, tok], [seq, d_model], axis=0, dict_len=vocab)
        x = b.add("Add", [x, b.const([seq, d_model], -wlim, wlim) if weights is None else b.const_data([seq, d_model], np.asarray(wq_["wpe"], dtype=np.int32))], [seq, d_model])
    else:
        x = b.add("Input", [], [seq, d_model])
    mask = b.const_data([seq, seq], np.tril(np.ones((seq, seq), dtype=np.int32)))
    maskb = b.add("Broadcast", [mask], [heads, seq, seq])
    neg = b.const_data([heads, seq, seq], np.full(heads * seq * seq, -(1 << (S + 3)) if mask_fill is None else mask_fill, dtype=np.int32))

    def full(dims, v):
        return b.const_data(dims, np.full(int(np.prod(dims)), v, dtype=np.int32))

    def norm(h):
        """LayerNorm: (h - mean) * rsqrt(mean((h - mean)^2) + eps) * w + bias"""
        if level < 1:
            return h
        if norm_kind == "rms":               # RMSNorm without learned parameters: h * rsqrt(mean(h^2) + eps)   (models/microgpt/gen.py:29-37)
            ms = b.add("MeanOfSquares", [h], [seq, 1], axes=[1], scale=S, count=d_model)
            ms = b.add("Add", [ms, full([seq, 1], 1)], [seq, 1])
            rs = b.add("Rsqrt", [ms], [seq, 1], scale=S)
            return b.add("Mul", [h, b.add("Broadcast", [rs], [seq, d_model])], [seq, d_model], scale=S)
        s = b.add("Sum", [h], [seq, 1], axes=[1])
        mean = b.add("ScalarConstDiv", [s], [seq, 1], divisor=d_model)
        c = b.add("Sub", [h, b.add("Broadcast", [mean], [seq, d_model])], [seq, d_model])
        var = b.add("MeanOfSquares", [c], [seq, 1], axes=[1], scale=S, count=d_model)
        var = b.add("Add", [var, full([seq, 1], 1)], [seq, 1])
        rs = b.add("Rsqrt", [var], [seq, 1], scale=S)
        y = b.add("Mul", [c, b.add("Broadcast", [rs], [seq, d_model])], [seq, d_model], scale=S)
        w = b.add("Broadcast", [lnw()], [seq, d_model])
        y = b.add("Mul", [y, w], [seq, d_model], scale=S)
        if not bias:                         # LayerNorm(ndim, bias=False): the weight only (models/nanoGPT/gen.py:28-37, config bias=False)
            return y
        return b.add("Add", [y, b.add("Broadcast", [b.const([d_model], -wlim, wlim)], [seq, d_model])], [seq, d_model])

    def gelu(f):
        """tanh-GELU: 0.5 f (1 + tanh(0.79788 (f + 0.044715 f^3)))"""
        if level < 2 or act == "relu":
            return b.add("ReLU", [f], [seq, ff])
        if cube_as_muls:                     # x * x * x as the model file writes it (two Mul nodes; torch.pow(x, 3) would be a Pow -> Cube)
            f3 = b.add("Mul", [b.add("Mul", [f, f], [seq, ff], scale=S), f], [seq, ff], scale=S)
        else:
            f3 = b.add("Cube", [f], [seq, ff], scale=S)
        u = b.add("Add", [f, b.add("Mul", [f3, full([seq, ff], qc(0.044715))], [seq, ff], scale=S)], [seq, ff])
        v = b.add("Mul", [u, full([seq, ff], qc(0.7978845608028654))], [seq, ff], scale=S)
        th = b.add("Tanh", [v], [seq, ff], scale=S)
        w = b.add("Add", [th, full([seq, ff], one)], [seq, ff])
        hx = b.add("Mul", [f, full([seq, ff], one // 2)], [seq, ff], scale=S)
        return b.add("Mul", [hx, w], [seq, ff], scale=S)

    if norm0:
        x = norm(x)
    for _ in range(layers):
        h = norm(x)
        # attention: q, k, v projections (one Einsum each: the tracer splits the fused qkv Gemm through Slice nodes)
        if fused_qkv:                        # c_attn: ONE MatMul into 3 d_model columns (padded to 4 d_model), then Split -> three Slice nodes
            if weights is None:
                wq = b.rng.integers(-wlim, wlim, size=(d_model, 4 * d_model)).astype(np.int32)
                wq[:, 3 * d_model:] = 0      # the loader's zero padding of 3 d_model -> the next power of two
                wqkv = b.const_data([d_model, 4 * d_model], wq)
            else:
                wqkv = mat(d_model, 4 * d_model)
            qkv = b.matmul(h, wqkv, seq, d_model, 4 * d_model, S)
            q, k, v = (b.add("Slice", [qkv], [seq, d_model], axis=1, start=j * d_model, end=(j + 1) * d_model) for j in range(3))
        else:
            q = b.matmul(h, mat(d_model, d_model), seq, d_model, d_model, S)
            k = b.matmul(h, mat(d_model, d_model), seq, d_model, d_model, S)
            v = b.matmul(h, mat(d_model, d_model), seq, d_model, d_model, S)
        qh = b.add("Reshape", [q], [seq, heads, hd])
        kh = b.add("Reshape", [k], [seq, heads, hd])
        vh = b.add("Reshape", [v], [seq, heads, hd])
        att = b.add("Einsum", [qh, kh], [heads, seq, seq], layout="mbk,nbk->bmn", scale=S, shape=[heads, seq, hd, seq])
        if att_scale:                        # (q k^T) * (1 / sqrt(head_dim)) as a fused-rescale Mul by a constant
            att = b.add("Mul", [att, full([heads, seq, seq], int(round(one / np.sqrt(hd))))], [heads, seq, seq], scale=S)
        att = b.add("Iff", [maskb, att, neg], [heads, seq, seq])
        att = b.add("SoftmaxLastAxis", [att], [heads, seq, seq], scale=S) if level >= 2 and S == 14 else b.add("ReLU", [att], [heads, seq, seq])
        y = b.add("Einsum", [att, vh], [seq, heads, hd], layout="bmk,kbn->mbn", scale=S, shape=[heads, seq, seq, hd])
        y = b.add("Reshape", [y], [seq, d_model])
        y = b.matmul(y, mat(d_model, d_model), seq, d_model, d_model, S)
        x = b.add("Add", [x, y], [seq, d_model])
        # MLP
        h = norm(x)
        f = b.matmul(h, mat(d_model, ff), seq, d_model, ff, S)
        if bias:
            f = b.add("Add", [f, b.add("Broadcast", [b.const([ff], -wlim, wlim)], [seq, ff])], [seq, ff])
        f = gelu(f)
        f = b.matmul(f, mat(ff, d_model), seq, ff, d_model, S)
        x = b.add("Add", [x, f], [seq, d_model])
    if final_norm:
        x = norm(x)
    if final_head:
        x = b.matmul(x, mat(d_model, vocab), seq, d_model, vocab, S)
    rng = np.random.default_rng(seed + 1)
    inputs = [rng.integers(0, vocab, size=seq).astype(np.int32)] if level >= 2 else [rng.integers(-one, one, size=seq * d_model).astype(np.int32)]
    if weights is not None and "tokens" in weights:
        inputs = [np.asarray(weights["tokens"], dtype=np.int32)]
    return b.nodes, [x], inputs


def nanogpt(level=2, seed=0, weights=None):
    """the shape of atlas-onnx-tracer/models/nanoGPT (gen.py:208-209: block 64, vocab 65 -> 128, 4 layers, 4 heads, n_embd 64, bias=False): LayerNorm
    with a weight and no bias, Linear layers without bias, the fused c_attn MatMul (64 x 192 -> 256) split three ways, the 1/sqrt(head_dim) score
    scale, masked scores filled with -10, tanh-GELU with x * x * x as two Mul nodes.  tests/test_model_shapes.py holds the operator counts of this
    graph against those of the model file (tests/golden/ref_model_ops.json)."""
    return transformer(layers=4, seq=64, d_model=64, heads=4, vocab=128, level=level, seed=seed, bias=False, att_scale=True, mask_fill=-10 * (1 << 14),
                       fused_qkv=True, cube_as_muls=True, weights=weights, round_consts=True)


def microgpt(level=2, seed=0, weights=None):
    """the shape of atlas-onnx-tracer/models/microgpt (gen.py:180-186; jolt-atlas-core/examples/microgpt.rs:20-31): vocab 32, n_embd 16, 4 heads,
    1 layer, block 16; RMSNorm without parameters (one right after the embedding, none before the head), ReLU MLP, no biases, the 1/sqrt(head_dim)
    score scale, masked scores filled with -10 (BASELINE config 1 shape)"""
    return transformer(layers=1, seq=16, d_model=16, heads=4, vocab=32, level=level, seed=seed, norm_kind="rms", act="relu", bias=False, norm0=True,
                       final_norm=False, att_scale=True, mask_fill=-10 * (1 << 14), weights=weights)


def _model_weights(key, n_ln):
    """tests/golden/ref_models.npz (tools/extract_ref_model.py): the model file's tensors quantized at 2^14 and padded, its example token ids"""
    import os
    Z = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "ref_models.npz"))
    lns = [Z[k] for k in sorted(Z.files) if k.startswith(key + "/ln")]
    return {"wte": Z[key + "/wte"], "wpe": Z[key + "/wpe"], "mats": [Z[k] for k in sorted(Z.files) if k.startswith(key + "/mat")],
            "ln": [lns[0]] * n_ln if lns else [], "tokens": Z[key + "/tokens"]}           # the exporter keeps ONE copy of the (identical) LayerNorm weights


def nanogpt_model():
    """nanogpt() over the tensors of atlas-onnx-tracer/models/nanoGPT/network.onnx and the example input of models/nanoGPT/input.json (BASELINE config 3)"""
    return nanogpt(weights=_model_weights("nanogpt", 9))


def microgpt_model():
    """microgpt() over the tensors of atlas-onnx-tracer/models/microgpt/network.onnx and its example input (BASELINE config 1)"""
    return microgpt(weights=_model_weights("microgpt", 0))


def gpt2_layer(level=2, seed=0, fused_qkv=True):
    # d_model 768 and 12 heads padded to 1024 / 16 (every dimension a power of two); the lm head is a 2^14-column slice; c_attn as in gpt2()
    return transformer(layers=1, seq=16, d_model=1024, heads=16, vocab=1 << 14, level=level, seed=seed, fused_qkv=fused_qkv)


def gpt2(level=2, seed=0, fused_qkv=True):
    # GPT-2 125M's operator list: 12 layers, d_model 768 -> 1024, 12 -> 16 heads, vocabulary 50257 -> 2^16 (embedding gather and lm head), seq 16.
    # c_attn is ONE MatMul 768 -> 2304 followed by a three-way Split, as the HF export jolt-atlas-core/examples/gpt2.rs:88-118 loads has it: padded
    # the way the loader pads (768 -> 1024 rows, 2304 -> 4096 columns), then three Slice nodes.  fused_qkv=False is the three 1024-wide projections
    # rounds 3-5 timed.
    return transformer(layers=12, seq=16, d_model=1024, heads=16, vocab=1 << 16, level=level, seed=seed, fused_qkv=fused_qkv)


def tiny(level=2, seed=0, layers=2):
    return transformer(layers=layers, seq=4, d_model=8, heads=2, vocab=16, level=level, seed=seed, mlp_mult=2)


def _node_rng():
    return np.random.default_rng(15)


def _rnd(rng, n):
    return rng.integers(-(1 << 14), 1 << 14, size=n, dtype=np.int64).astype(np.int32)


def node_einsum():
    """one operator at the size bench.py times: the GPT-2 MLP projection 16 x 1024 . 1024 x 4096 at scale 2^14 (T = 2^16, k = 1024)"""
    rng = _node_rng()
    w = _rnd(rng, 1024 * 4096)
    return [{"idx": 0, "op": "Input", "inputs": [], "dims": [16, 1024]},
            {"idx": 1, "op": "Constant", "inputs": [], "dims": [1024, 4096], "data": w},
            {"idx": 2, "op": "Einsum", "inputs": [0, 1], "dims": [16, 4096], "layout": "mk,kn->mn", "scale": 14, "shape": [16, 1024, 4096]}], [2], [_rnd(rng, 16 * 1024)]


def node_relu():
    rng = _node_rng()
    return [{"idx": 0, "op": "Input", "inputs": [], "dims": [16, 4096]}, {"idx": 1, "op": "ReLU", "inputs": [0], "dims": [16, 4096]}], [1], [_rnd(rng, 1 << 16)]


def node_add():
    rng = _node_rng()
    c = _rnd(rng, 1 << 16)
    return [{"idx": 0, "op": "Input", "inputs": [], "dims": [16, 4096]},
            {"idx": 1, "op": "Constant", "inputs": [], "dims": [16, 4096], "data": c},
            {"idx": 2, "op": "Add", "inputs": [0, 1], "dims": [16, 4096]}], [2], [_rnd(rng, 1 << 16)]


def node_mul():
    rng = _node_rng()
    c = _rnd(rng, 1 << 16)
    return [{"idx": 0, "op": "Input", "inputs": [], "dims": [16, 4096]},
            {"idx": 1, "op": "Constant", "inputs": [], "dims": [16, 4096], "data": c},
            {"idx": 2, "op": "Mul", "inputs": [0, 1], "dims": [16, 4096], "scale": 14}], [2], [_rnd(rng, 1 << 16)]


NO_COMMIT = {"Input", "Constant", "Identity", "Reshape", "MoveAxis", "Broadcast", "Slice", "Iff", "And", "Concat"}


def max_vars(nodes):
    """log2 of the largest committed polynomial (AtlasSharedPreprocessing::max_num_vars): a one-hot chunk has 16 x T coefficients, T = the
    node's (padded) element count (Gather: its index count; GatherSmall's one polynomial has dict_len x T); dense advice polynomials have T coefficients"""
    byidx = {nd["idx"]: nd for nd in nodes}
    best = 4
    for nd in nodes:
        if nd["op"] in NO_COMMIT:
            continue
        T = int(np.prod(byidx[nd["inputs"][1]]["dims"])) if nd["op"] in ("GatherLarge", "GatherSmall") else int(np.prod(nd["dims"]))
        lk = int(np.log2(byidx[nd["inputs"][0]]["dims"][0])) if nd["op"] == "GatherSmall" else 4        # GatherRa: all of the dictionary's address bits
        best = max(best, lk + int(np.log2(max(T, 1))))
    return best
