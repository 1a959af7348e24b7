#!/usr/bin/env python3
"""The tensors of the reference's model files and their stored example input / output, as data fixtures:

    atlas-onnx-tracer/models/{microgpt,nanoGPT}/network.onnx   initializers (float32) -> quantized i32 at MODEL_SCALE = 14 by the loader's rule
                                                               (quantize_float, utils/quantize.rs:137-185: round half away from zero, a non-zero float that
                                                               rounds to 0 becomes +-1), padded with zeros to powers of two per dimension (model/load.rs:505-559)
    atlas-onnx-tracer/models/{microgpt,nanoGPT}/input.json     the example token ids and the float logits the exporting script stored beside the model

    python tools/extract_ref_model.py      # -> tests/golden/ref_models.npz   (build container; the reference is not on the GPU box)

tests/test_model_fidelity.py builds the graphs of tools/build_graphs.py over these tensors, executes them (oracle executor on the CPU, atlas_graph_trace
on the GPU) and compares the dequantized logits with the stored float output."""
import json
import os
import struct
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from onnx_op_histogram import fields, packed_varints   # noqa: E402

MODELS = "/root/reference/atlas-onnx-tracer/models"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "ref_models.npz")
S = 14


def initializers(path):
    data = open(path, "rb").read()
    graph = next(v for f, wt, v in fields(data) if f == 7 and wt == 2)
    out = []
    for f, wt, v in fields(graph):
        if f != 5 or wt != 2:
            continue
        dims, name, dtype, raw, floats = [], "", 0, None, []
        for ff, w, x in fields(v):
            if ff == 1:
                dims += packed_varints(x) if w == 2 else [x]
            elif ff == 2:
                dtype = x
            elif ff == 4:
                floats += list(struct.unpack("<%df" % (len(x) // 4), x)) if w == 2 else [struct.unpack("<f", x)[0]]
            elif ff == 8 and w == 2:
                name = x.decode()
            elif ff == 9 and w == 2:
                raw = x
        if dtype != 1:                                   # FLOAT only: the int64 shape constants are not weights
            continue
        a = np.frombuffer(raw, dtype="<f4") if raw is not None else np.array(floats, dtype=np.float32)
        out.append((name, a.reshape(dims) if dims else a.reshape(())))
    return out


def quantize(a):
    """quantize_float at scale 14 (no value of these models is near the i32 range)"""
    f = a.astype(np.float64) * (1 << S)
    q = (np.floor(np.abs(f) + 0.5) * np.sign(f)).astype(np.int64)
    q = np.where((q == 0) & (a != 0), np.sign(a).astype(np.int64), q)
    return q.astype(np.int32)


def pad2(a):
    dims = [1 << max(int(d) - 1, 0).bit_length() for d in a.shape]
    out = np.zeros(dims, dtype=a.dtype)
    out[tuple(slice(0, d) for d in a.shape)] = a
    return out


def _softmax_masked(att, fill=-10.0):
    n = att.shape[-1]
    att = np.where(np.tril(np.ones((n, n))) == 0, fill, att)
    att = np.exp(att - att.max(-1, keepdims=True))
    return att / att.sum(-1, keepdims=True)


def nanogpt_forward(wte, wpe, lnw, mats, tokens, heads=4):
    """models/nanoGPT/gen.py in float64 over the file's tensors"""
    import math
    m = [a.astype(np.float64) for a in mats]
    lnw = lnw.astype(np.float64)
    T, C = len(tokens), wte.shape[1]
    hs = C // heads

    def ln(x):
        mu = x.mean(-1, keepdims=True); v = ((x - mu) ** 2).mean(-1, keepdims=True)
        return (x - mu) / np.sqrt(v + 1e-5) * lnw
    x = wte.astype(np.float64)[tokens] + wpe.astype(np.float64)[:T]
    mi = 0
    for _ in range((len(m) - 1) // 4):
        qkv = ln(x) @ m[mi]; mi += 1
        q, k, v = (qkv[:, j * C:(j + 1) * C].reshape(T, heads, hs).transpose(1, 0, 2) for j in range(3))
        y = (_softmax_masked(q @ k.transpose(0, 2, 1) / math.sqrt(hs)) @ v).transpose(1, 0, 2).reshape(T, C)
        x = x + y @ m[mi]; mi += 1
        f = ln(x) @ m[mi]; mi += 1
        f = 0.5 * f * (1 + np.tanh(math.sqrt(2 / math.pi) * (f + 0.044715 * f ** 3)))
        x = x + f @ m[mi]; mi += 1
    return ln(x) @ m[mi]


def microgpt_forward(wte, wpe, mats, tokens, heads=4):
    """models/microgpt/gen.py in float64 over the file's tensors"""
    import math
    m = [a.astype(np.float64) for a in mats]
    T, C = len(tokens), wte.shape[1]
    hs = C // heads
    rms = lambda x: x / np.sqrt((x * x).mean(-1, keepdims=True) + 1e-5)
    x = rms(wte.astype(np.float64)[tokens] + wpe.astype(np.float64)[:T])
    h = rms(x)
    q, k, v = ((h @ m[j]).reshape(T, heads, hs).transpose(1, 0, 2) for j in range(3))
    y = (_softmax_masked(q @ k.transpose(0, 2, 1) / math.sqrt(hs)) @ v).transpose(1, 0, 2).reshape(T, C)
    x = x + y @ m[3]
    x = x + np.maximum(rms(x) @ m[4], 0) @ m[5]
    return x @ m[6]


def main():
    out = {}
    for model, key in (("microgpt", "microgpt"), ("nanoGPT", "nanogpt")):
        init = initializers(os.path.join(MODELS, model, "network.onnx"))
        byname = dict(init)
        wte = next(a for n, a in init if n.endswith("wte.weight"))
        wpe = next(a for n, a in init if n.endswith("wpe.weight"))
        mats = [a for n, a in init if n.startswith("onnx::MatMul")]                      # in order of use (the exporter numbers them so)
        lns = [a for n, a in init if ".ln_" in n and n.endswith(".weight")]                # ln_1, ln_2 per layer, then ln_f
        ex = json.load(open(os.path.join(MODELS, model, "input.json")))
        tokens = np.array(ex["input_data"][0], dtype=np.float64).astype(np.int32)
        out[f"{key}/tokens"] = tokens
        out[f"{key}/logits_f32"] = np.array(ex["output_data"][0], dtype=np.float32)
        out[f"{key}/wte"] = pad2(quantize(wte))
        out[f"{key}/wpe"] = pad2(quantize(wpe[:len(tokens)]))
        for i, m in enumerate(mats):
            out[f"{key}/mat{i:02d}"] = pad2(quantize(m))
        for i, v in enumerate(lns):
            out[f"{key}/ln{i:02d}"] = pad2(quantize(v))
        if key == "nanogpt":
            # The logits stored in models/nanoGPT/input.json are NOT those of this network.onnx on this input (a float64 forward pass over the
            # file's own tensors correlates 0.32 with them; microgpt's stored logits agree with its file to 1e-6): the expected output of the
            # nanoGPT fixture is that forward pass (standard GPT: LayerNorm weight only, fused c_attn, causal mask filled with -10, tanh-GELU).
            out[f"{key}/logits_float_forward"] = nanogpt_forward(wte, wpe, lns[0], mats, tokens).astype(np.float32)
            c = np.corrcoef(out[f"{key}/logits_float_forward"].ravel(), out[f"{key}/logits_f32"].ravel())[0, 1]
            print("   nanoGPT: float forward over the file's tensors vs the stored output_data: correlation %.3f" % c)
        else:
            out[f"{key}/logits_float_forward"] = microgpt_forward(wte, wpe, mats, tokens).astype(np.float32)
            print("   microgpt: float forward vs the stored output_data: max abs difference %.2e" % np.abs(out[f"{key}/logits_float_forward"].ravel() - out[f"{key}/logits_f32"]).max())
        print(model, "initializers", [(n, list(a.shape)) for n, a in init if a.ndim][:6], "...", len(mats), "matrices", len(lns), "LayerNorm weights", "tokens", len(tokens),
              "logits", out[f"{key}/logits_f32"].shape, [n for n in byname if "bias" in n][:3])
    np.savez_compressed(OUT, **out)
    print(OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
