"""GPU: ONNXProof::prove over whole model graphs (atlas_prove_graph: trace on the device, witness commitments, output claim, the
reverse node loop with NodeEvalReduction, every operator's composition, the reduced opening proof, the serialized ONNXProof)
against the same graph proved on the CPU by oracle/graph.py — proof bytes and final transcript state, byte for byte."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _const(idx, rng, dims, lo=-64, hi=64):
    return {"idx": idx, "op": "Constant", "inputs": [], "dims": list(dims), "data": rng.integers(lo, hi, size=int(np.prod(dims))).astype(np.int32)}


def mlp_graph(rng, m=4, k=8, h=16, S=5):
    return [
        {"idx": 0, "op": "Input", "inputs": [], "dims": [m, k]},
        _const(1, rng, [k, h]),
        {"idx": 2, "op": "Einsum", "inputs": [0, 1], "dims": [m, h], "layout": "mk,kn->mn", "scale": S, "shape": [m, k, h]},
        _const(3, rng, [m, h]),
        {"idx": 4, "op": "Add", "inputs": [2, 3], "dims": [m, h]},
        {"idx": 5, "op": "ReLU", "inputs": [4], "dims": [m, h]},
        _const(6, rng, [h, k]),
        {"idx": 7, "op": "Einsum", "inputs": [5, 6], "dims": [m, k], "layout": "mk,kn->mn", "scale": S, "shape": [m, h, k]},
        {"idx": 8, "op": "Square", "inputs": [7], "dims": [m, k], "scale": S},
        {"idx": 9, "op": "Mul", "inputs": [8, 0], "dims": [m, k], "scale": S},
        {"idx": 10, "op": "Sub", "inputs": [9, 0], "dims": [m, k]},
    ], [10], [rng.integers(-64, 64, size=m * k).astype(np.int32)]


def shape_graph(rng, t=4, b=2, hd=4, S=4):
    """the attention-shaped slice: reshape to heads, q k^T (mbk,nbk->bmn), mask select, att v (bmk,kbn->mbn), residual, cube"""
    c = b * hd
    mask = np.tril(np.ones((t, t), dtype=np.int32))
    return [
        {"idx": 0, "op": "Input", "inputs": [], "dims": [t, c]},
        {"idx": 1, "op": "Reshape", "inputs": [0], "dims": [t, b, hd]},
        {"idx": 2, "op": "Reshape", "inputs": [0], "dims": [t, b, hd]},
        {"idx": 3, "op": "Einsum", "inputs": [1, 2], "dims": [b, t, t], "layout": "mbk,nbk->bmn", "scale": S, "shape": [b, t, hd, t]},
        {"idx": 4, "op": "Constant", "inputs": [], "dims": [t, t], "data": mask.reshape(-1)},
        {"idx": 5, "op": "Broadcast", "inputs": [4], "dims": [b, t, t]},
        _const(6, rng, [b, t, t], -8, 8),
        {"idx": 7, "op": "Iff", "inputs": [5, 3, 6], "dims": [b, t, t]},
        {"idx": 8, "op": "Einsum", "inputs": [7, 1], "dims": [t, b, hd], "layout": "bmk,kbn->mbn", "scale": S, "shape": [b, t, t, hd]},
        {"idx": 9, "op": "Reshape", "inputs": [8], "dims": [t, c]},
        {"idx": 10, "op": "Add", "inputs": [9, 0], "dims": [t, c]},
        {"idx": 11, "op": "MoveAxis", "inputs": [10], "dims": [c, t], "source": 0, "destination": 1},
        {"idx": 12, "op": "Cube", "inputs": [11], "dims": [c, t], "scale": S},
        {"idx": 13, "op": "Identity", "inputs": [12], "dims": [c, t]},
        {"idx": 14, "op": "And", "inputs": [5, 5], "dims": [b, t, t]},
        {"idx": 15, "op": "Einsum", "inputs": [14, 2], "dims": [t, b, hd], "layout": "bmk,kbn->mbn", "scale": S, "shape": [b, t, t, hd]},
        {"idx": 16, "op": "Reshape", "inputs": [15], "dims": [c, t]},
        {"idx": 17, "op": "Sub", "inputs": [13, 16], "dims": [c, t]},
    ], [17], [rng.integers(-32, 32, size=t * c).astype(np.int32)]


def norm_graph(rng, t=4, c=8, S=6):
    """the LayerNorm-shaped slice: mean (Sum + ScalarConstDiv), centring (Broadcast + Sub), MeanOfSquares, Add eps, Rsqrt, Broadcast,
    Mul; then a Div by a positive tensor and a Slice"""
    one = 1 << S
    return [
        {"idx": 0, "op": "Input", "inputs": [], "dims": [t, c]},
        {"idx": 1, "op": "Sum", "inputs": [0], "dims": [t, 1], "axes": [1]},
        {"idx": 2, "op": "ScalarConstDiv", "inputs": [1], "dims": [t, 1], "divisor": c},
        {"idx": 3, "op": "Broadcast", "inputs": [2], "dims": [t, c]},
        {"idx": 4, "op": "Sub", "inputs": [0, 3], "dims": [t, c]},
        {"idx": 5, "op": "MeanOfSquares", "inputs": [4], "dims": [t, 1], "axes": [1], "scale": S, "count": c},
        {"idx": 6, "op": "Constant", "inputs": [], "dims": [t, 1], "data": np.full(t, 3, dtype=np.int32)},
        {"idx": 7, "op": "Add", "inputs": [5, 6], "dims": [t, 1]},
        {"idx": 8, "op": "Rsqrt", "inputs": [7], "dims": [t, 1], "scale": S},
        {"idx": 9, "op": "Broadcast", "inputs": [8], "dims": [t, c]},
        {"idx": 10, "op": "Mul", "inputs": [4, 9], "dims": [t, c], "scale": S},
        {"idx": 11, "op": "Constant", "inputs": [], "dims": [t, c], "data": rng.integers(1, 9, size=t * c).astype(np.int32)},
        {"idx": 12, "op": "Div", "inputs": [10, 11], "dims": [t, c]},
        {"idx": 13, "op": "Slice", "inputs": [12], "dims": [t, c // 2], "axis": 1, "start": c // 2, "end": c},
        {"idx": 14, "op": "Sum", "inputs": [13], "dims": [1, c // 2], "axes": [0]},
    ], [14], [rng.integers(-4 * one, 4 * one, size=t * c).astype(np.int32)]


def act_graph(rng, n=4, d=4, v=8, S=14):
    """embedding + activation: GatherLarge of a [v][d] dictionary, Mul by a constant, Tanh (scale 14: the prover's table), Add"""
    dict_ = rng.integers(-(1 << 16), 1 << 16, size=v * d).astype(np.int32)
    big = rng.integers(-(1 << 19), 1 << 19, size=n * d).astype(np.int32)
    return [
        {"idx": 0, "op": "Input", "inputs": [], "dims": [n]},
        {"idx": 1, "op": "Constant", "inputs": [], "dims": [v, d], "data": dict_},
        {"idx": 2, "op": "GatherLarge", "inputs": [1, 0], "dims": [n, d], "axis": 0, "dict_len": v},
        {"idx": 3, "op": "Constant", "inputs": [], "dims": [n, d], "data": big},
        {"idx": 4, "op": "Add", "inputs": [2, 3], "dims": [n, d]},
        {"idx": 5, "op": "Tanh", "inputs": [4], "dims": [n, d], "scale": S},
        {"idx": 6, "op": "Add", "inputs": [5, 2], "dims": [n, d]},
    ], [6], [rng.integers(0, v, size=n).astype(np.int32)]


def softmax_graph(rng, b=2, t=4, n=8, S=14):
    """SoftmaxLastAxis over [b][t][n] scores (scale 14: the clamp table of the prover), with masked entries far below the row maximum
    (beyond the exp tables: the saturating clamp) and ties for the maximum; followed by an einsum that consumes the probabilities"""
    x = rng.integers(-(1 << 16), 1 << 16, size=b * t * n).astype(np.int32).reshape(b, t, n)
    x[0, 1, 3:] = -(1 << 19); x[1, 2, 5] = -(1 << 17); x[1, 0, :] = 777; x[0, 3, 2] = x[0, 3].max()
    return [
        {"idx": 0, "op": "Input", "inputs": [], "dims": [b, t, n]},
        {"idx": 1, "op": "SoftmaxLastAxis", "inputs": [0], "dims": [b, t, n], "scale": S},
        _const(2, rng, [n, b, 4], -(1 << 12), 1 << 12),
        {"idx": 3, "op": "Einsum", "inputs": [1, 2], "dims": [t, b, 4], "layout": "bmk,kbn->mbn", "scale": S, "shape": [b, t, n, 4]},
    ], [3], [x.reshape(-1)]


def erf_sigmoid_graph(rng, n=4, d=8, S=14):
    """the other two small-table activations (ops/erf.rs, sigmoid.rs = prove_clamped_activation with their own tables): Erf of a wide-range
    input (beyond the clamp on both sides), Sigmoid of a product"""
    x = rng.integers(-(1 << 18), 1 << 18, size=n * d).astype(np.int32)
    x[:4] = [0, -1, (1 << 17) - 1, -(1 << 17)]
    return [
        {"idx": 0, "op": "Input", "inputs": [], "dims": [n, d]},
        {"idx": 1, "op": "Erf", "inputs": [0], "dims": [n, d], "scale": S},
        _const(2, rng, [n, d], -(1 << 15), 1 << 15),
        {"idx": 3, "op": "Mul", "inputs": [1, 2], "dims": [n, d], "scale": S},
        {"idx": 4, "op": "Sigmoid", "inputs": [3], "dims": [n, d], "scale": S},
        {"idx": 5, "op": "Add", "inputs": [4, 1], "dims": [n, d]},
    ], [5], [x]


def small_ops_graph(rng, n=4, d=8):
    """Neg and IsNan (no sumcheck: the operand opened at the reduced point) and the ONNX Clamp (ops/clamp.rs: ClampTable<32>, bound 2^9) on an
    input that exceeds the bound on both sides"""
    x = rng.integers(-(1 << 12), 1 << 12, size=n * d).astype(np.int32)
    x[:6] = [0, -1, 511, 512, -512, -513]
    return [
        {"idx": 0, "op": "Input", "inputs": [], "dims": [n, d]},
        {"idx": 1, "op": "Neg", "inputs": [0], "dims": [n, d]},
        {"idx": 2, "op": "Clamp", "inputs": [0], "dims": [n, d], "bound_log": 9},
        {"idx": 3, "op": "Add", "inputs": [1, 2], "dims": [n, d]},
        {"idx": 4, "op": "IsNan", "inputs": [0], "dims": [n, d]},
        {"idx": 5, "op": "Add", "inputs": [3, 4], "dims": [n, d]},
    ], [5], [x]


def gather_small_graph(rng, n=8, d=4, v=32):
    """the embedding as the tracer emits it for vocabularies of at most 2^16 words (handlers/index.rs:33-45): GatherSmall — ONE committed
    one-hot polynomial over all v addresses, HammingBooleanity + Booleanity (d = 1) batched, HammingWeight on its own — then a Mul"""
    dict_ = rng.integers(-(1 << 14), 1 << 14, size=v * d).astype(np.int32)
    return [
        {"idx": 0, "op": "Input", "inputs": [], "dims": [n]},
        {"idx": 1, "op": "Constant", "inputs": [], "dims": [v, d], "data": dict_},
        {"idx": 2, "op": "GatherSmall", "inputs": [1, 0], "dims": [n, d], "axis": 0, "dict_len": v},
        _const(3, rng, [n, d], -(1 << 14), 1 << 14),
        {"idx": 4, "op": "Mul", "inputs": [2, 3], "dims": [n, d], "scale": 14},
    ], [4], [rng.integers(0, v, size=n).astype(np.int32)]


def scalar_graph(rng):
    """scalar nodes (one element: `is_scalar`, clamp_lookups/mod.rs:67): no lookups, the accumulation opens in the clear and the verifier
    recomputes the clamp — Sum down to one element, Add, fused-rescale Mul, Square and Cube of scalars, Sub, then back to a vector"""
    return [
        {"idx": 0, "op": "Input", "inputs": [], "dims": [2, 2]},
        {"idx": 1, "op": "Sum", "inputs": [0], "dims": [2, 1], "axes": [1]},
        {"idx": 2, "op": "Sum", "inputs": [1], "dims": [1, 1], "axes": [0]},
        _const(3, rng, [1, 1]),
        {"idx": 4, "op": "Add", "inputs": [2, 3], "dims": [1, 1]},
        _const(5, rng, [1, 1]),
        {"idx": 6, "op": "Mul", "inputs": [4, 5], "dims": [1, 1], "scale": 3},
        {"idx": 7, "op": "Square", "inputs": [6], "dims": [1, 1], "scale": 4},
        {"idx": 8, "op": "Cube", "inputs": [6], "dims": [1, 1], "scale": 5},
        {"idx": 9, "op": "Sub", "inputs": [8, 7], "dims": [1, 1]},
        {"idx": 10, "op": "Broadcast", "inputs": [9], "dims": [1, 2]},
        _const(11, rng, [1, 2]),
        {"idx": 12, "op": "Add", "inputs": [10, 11], "dims": [1, 2]},
    ], [12], [rng.integers(-100, 100, size=4).astype(np.int32)]


def trig_graph(rng):
    """Sin and Cos by neural teleportation (ops/sin.rs, cos.rs): inputs around multiples of the period modulus (the reference's own boundary
    cases, ops/sin.rs test_sin_periodic_boundary_inputs) and random ones; Cos of the Sin output; an Add on top"""
    M = 2470649
    x = rng.integers(-50000, 50000, size=16).astype(np.int32)
    x[:8] = [-M - 1, -M, -1, 0, 1, M - 1, M, M + 1]
    return [
        {"idx": 0, "op": "Input", "inputs": [], "dims": [2, 8]},
        {"idx": 1, "op": "Sin", "inputs": [0], "dims": [2, 8], "scale": 14},
        {"idx": 2, "op": "Cos", "inputs": [1], "dims": [2, 8], "scale": 14},
        {"idx": 3, "op": "Cos", "inputs": [0], "dims": [2, 8], "scale": 14},
        {"idx": 4, "op": "Add", "inputs": [2, 3], "dims": [2, 8]},
    ], [4], [x]


def concat_graph(rng):
    """Concat (ops/concat.rs) along the last axis of three operands of unequal size (the smaller ones are repeated over the low variables of the
    largest one's hypercube), an Add over the result, then a second Concat along axis 0 whose output carries the output claim"""
    return [
        {"idx": 0, "op": "Input", "inputs": [], "dims": [4, 2]},
        _const(1, rng, [4, 2]),
        _const(2, rng, [4, 4]),
        {"idx": 3, "op": "Concat", "inputs": [0, 1, 2], "dims": [4, 8], "axis": 1},
        _const(4, rng, [4, 8]),
        {"idx": 5, "op": "Add", "inputs": [3, 4], "dims": [4, 8]},
        {"idx": 6, "op": "Concat", "inputs": [5, 3], "dims": [8, 8], "axis": 0},
    ], [6], [rng.integers(-64, 64, size=8).astype(np.int32)]


def _max_vars(nodes):
    # the largest committed polynomial is a one-hot chunk: K = 16 addresses x T cycles (GatherSmall: dict_len addresses x the index count)
    by_idx = {nd["idx"]: nd for nd in nodes}
    gs = [int(np.log2(nd["dict_len"])) + int(np.log2(np.prod(by_idx[nd["inputs"][1]]["dims"]))) for nd in nodes if nd["op"] == "GatherSmall"]
    return max([4 + max(int(np.log2(np.prod(nd["dims"]))) for nd in nodes)] + gs)


def toy_transformer(rng):
    """two decoder layers (tools/build_graphs.py: embedding gather, LayerNorm, attention einsums, tanh-GELU MLP, lm head), seq 4, d_model 8"""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import build_graphs as BG
    return BG.tiny(layers=2)


# (the 2-layer toy transformer is proved against the committed oracle result: tests/test_gpu_graph_golden.py, "tiny2" — 24 s of oracle time here)
@pytest.mark.parametrize("builder,seed", [(mlp_graph, 1), (shape_graph, 2), (norm_graph, 3), (act_graph, 4), (softmax_graph, 6), (concat_graph, 7), (erf_sigmoid_graph, 8), (small_ops_graph, 9), (gather_small_graph, 10), (scalar_graph, 11), (trig_graph, 12)])
def test_graph_proof_matches_oracle(atlas, builder, seed):
    from oracle import graph as OG, orc
    from jolt_atlas_amd import graph as GG
    rng = np.random.default_rng(seed)
    nodes, outputs, inputs = builder(rng)
    nv = _max_vars(nodes)
    tau = orc.random_fr(1, 0x51250001)[0]
    srs_h = orc.srs_powers(tau, 1 << nv)
    srs = atlas.SRS.generate(tau, 1 << nv)
    # ---- oracle
    P = OG.Prover(nodes, outputs, srs_h)
    want = P.prove(inputs)
    # ---- device
    G = GG.Graph(nodes, outputs)
    G.trace(inputs)
    for nd in nodes:
        assert np.array_equal(G.node_output(nd["idx"]), P.trace[nd["idx"]]), f"trace of node {nd['idx']} ({nd['op']})"
    got, state, tm = G.prove(srs, inputs)
    assert state == P.t.state(), "final transcript state"
    assert got == want, "ONNXProof bytes"
    assert tm["n_nodes"] == len(nodes) and tm["n_committed"] == len(P.committed)
    G.free(); srs.free()


@pytest.mark.parametrize("builder,seed", [(mlp_graph, 1), (shape_graph, 2), (norm_graph, 3), (act_graph, 4), (toy_transformer, 5), (softmax_graph, 6), (concat_graph, 7), (erf_sigmoid_graph, 8), (small_ops_graph, 9), (gather_small_graph, 10), (scalar_graph, 11), (trig_graph, 12)])
def test_graph_proof_is_accepted_by_the_verifier(atlas, builder, seed):
    """ONNXProof::verify (atlas_verify_graph: opening claims from the proof, the node loop's verifier instances, the opening-reduction
    sumcheck, the joint commitment, HyperKZG::verify through the pairing) accepts the device's proof with the prover's final transcript
    state, and rejects a wrong output tensor, a tampered opening claim, a tampered round polynomial and a tampered commitment."""
    from oracle import orc
    from jolt_atlas_amd import graph as GG
    rng = np.random.default_rng(seed)
    nodes, outputs, inputs = builder(rng)
    nv = _max_vars(nodes)
    tau = orc.random_fr(1, 0x51250001)[0]
    srs = atlas.SRS.generate(tau, 1 << nv)
    vk = atlas.HyperKZG.vk_from_trapdoor(tau, srs.download(0, 1)[0])
    G = GG.Graph(nodes, outputs)
    proof, state, _ = G.prove(srs, inputs)
    out = G.node_output(outputs[0])
    V = GG.Graph(nodes, outputs)                         # a verifier's graph: never traced
    ok, vstate = V.verify(vk, inputs, out, proof)
    assert ok and vstate == state
    bad_out = out.copy(); bad_out[0] += 1
    assert not V.verify(vk, inputs, bad_out, proof)[0]
    n_open = int.from_bytes(proof[:8], "little")
    assert n_open > 10
    offs = [8 + 13 + 5,                                  # inside the first opening claim
            len(proof) // 2,                             # somewhere in the sumcheck proofs / commitments
            len(proof) - 40]                             # the last HyperKZG evaluation
    offs += [int(x) for x in np.random.default_rng(seed + 100).integers(8, len(proof), 6)]
    for off in offs:
        bad = bytearray(proof); bad[off] ^= 1
        try:
            assert not V.verify(vk, inputs, out, bytes(bad))[0], off
        except atlas.AtlasError:
            pass                                         # a flipped bit that breaks the ark encoding is rejected as malformed
    G.free(); V.free(); srs.free()
