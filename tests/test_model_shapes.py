"""The graphs `tools/build_graphs.py` emits for the reference's model files, against what the files hold
(tests/golden/ref_model_ops.json: operator types and initializer shapes of atlas-onnx-tracer/models/*/network.onnx, data only, read
by tools/onnx_op_histogram.py in the build container).

The loader (`Model::load_onnx_model`) sends the file through tract's typing + decluttering before its handlers map each tract operator
onto the prover's vocabulary (atlas-onnx-tracer/src/node/handlers/*.rs); tract is not in this image, so the node list is the builder's
reading of that pipeline.  These tests hold the reading to the counts that survive it: one Einsum per MatMul, one SoftmaxLastAxis per
Softmax, one Tanh / ReLU per activation, one Iff per Where, one Rsqrt per Sqrt (Sqrt + Div -> Rsqrt + Mul, handlers/arith.rs), the
ReduceMean of a LayerNorm as Sum + ScalarConstDiv and Pow + ReduceMean as MeanOfSquares (handlers/reduce.rs:40-52), the weights' shapes."""
import json
import os
import sys
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
REF = json.load(open(os.path.join(ROOT, "tests", "golden", "ref_model_ops.json")))["models"]


def _ops(nodes):
    return Counter(n["op"] for n in nodes)


def _pow2(x):
    return 1 << (x - 1).bit_length()


def test_nanogpt_graph_follows_the_model_file():
    import build_graphs as BG
    nodes, outputs, inputs = BG.nanogpt()
    ours, ref = _ops(nodes), REF["nanoGPT"]["ops"]
    assert ours["Einsum"] == ref["MatMul"] == 25                      # c_attn (fused q, k, v), q k^T, att v, c_proj, fc, proj per layer + lm_head
    assert ours["SoftmaxLastAxis"] == ref["Softmax"] == 4
    assert ours["Tanh"] == ref["Tanh"] == 4
    assert ours["Iff"] == ref["Where"] == 4
    assert ours["Rsqrt"] == ref["Sqrt"] == 9                          # 2 LayerNorms per layer + ln_f
    assert ours["MeanOfSquares"] == ref["Pow"] == 9                   # Pow(2) + ReduceMean -> Reduce<MeanOfSquares>
    assert ours["Sum"] + ours["MeanOfSquares"] == ref["ReduceMean"] == 18
    assert ours["ScalarConstDiv"] == ours["Sum"] == 9                 # the mean: Sum, then the division by the element count
    assert ours["Sub"] == 9 and ref["Sub"] in (9, 10)                 # x - mean per LayerNorm (the file has one more in its shape arithmetic)
    # Mul: 37 in the file (per layer: the LayerNorm weights x 2, the score scale, six in the tanh-GELU with x * x * x as two; ln_f's weight) + one per
    # LayerNorm for (x - mean) / sqrt(var + eps), a Div in the file, Rsqrt + Mul here
    assert ours["Mul"] == ref["Mul"] + ours["Rsqrt"] == 46
    assert "Cube" not in ours and "ReLU" not in ours
    assert ours["Add"] in (ref["Add"], ref["Add"] - 1)                # bias=False: eps x 2, residual x 2, two in the GELU per layer; embeddings; ln_f
    assert ours["Slice"] == 3 * ref["Split"] == 12                    # Split of the fused projection three ways
    assert ours["GatherSmall"] == 1 and len(inputs) == 1 and len(inputs[0]) == 64
    dims = REF["nanoGPT"]["initializer_dims"]
    consts = Counter(tuple(n["dims"]) for n in nodes if n["op"] == "Constant")
    assert dims["transformer.wte.weight"] == [65, 64] and consts[(128, 64)] == 1          # vocabulary 65 -> 128
    assert dims["transformer.wpe.weight"] == [64, 64]
    mm = [d for k, d in dims.items() if k.startswith("onnx::MatMul")]
    assert sorted(mm) == sorted([[64, 192], [64, 64], [64, 256], [256, 64]] * 4 + [[64, 65]])
    for d in mm:                                                      # every weight matrix of the file is a Constant of the padded shape here
        assert consts[tuple(_pow2(x) for x in d)] >= 1, d
    assert consts[(64, 256)] >= 8                                     # c_attn (192 -> 256) and fc per layer (+ the GELU constants over [seq][ff])


def test_microgpt_graph_follows_the_model_file():
    import build_graphs as BG
    nodes, outputs, inputs = BG.microgpt()
    ours, ref = _ops(nodes), REF["microgpt"]["ops"]
    assert ours["Einsum"] == ref["MatMul"] == 9                       # wq, wk, wv, q k^T, att v, wo, fc1, fc2, lm_head
    assert ours["SoftmaxLastAxis"] == ref["Softmax"] == 1
    assert ours["ReLU"] == ref["Relu"] == 1
    assert ours["Iff"] == ref["Where"] == 1
    assert ours["MeanOfSquares"] == ref["ReduceMean"] == 3            # RMSNorm: mean(x * x), x * x folded into Reduce<MeanOfSquares>
    assert ours["Rsqrt"] == ref["Sqrt"] == 3
    assert ours["Mul"] == ref["Mul"] - ours["MeanOfSquares"] == 4     # the file's 7 = 3 (x * x) + 3 (x * rsqrt) + the score scale
    assert ours["Add"] == 6 and ref["Add"] == 7                       # embeddings, eps x 3, residual x 2 (the file has one more in its shape arithmetic)
    dims = REF["microgpt"]["initializer_dims"]
    assert dims["wte.weight"] == [32, 16] and dims["wpe.weight"] == [16, 16]
    mm = sorted(d for k, d in dims.items() if k.startswith("onnx::MatMul"))
    assert mm == sorted([[16, 16]] * 4 + [[16, 64], [64, 16], [16, 32]])
    consts = Counter(tuple(n["dims"]) for n in nodes if n["op"] == "Constant")
    for d in mm:
        assert consts[tuple(d)] >= 1, d
    assert len(inputs[0]) == 16
