#!/usr/bin/env python3
"""Operator histograms of the reference's ONNX model files (atlas-onnx-tracer/models/*/network.onnx), read with a minimal protobuf
wire-format walker (no `onnx` package in this image): ModelProto.graph (field 7) -> GraphProto.node (1) -> NodeProto.op_type (4), plus the
initializers' dims (GraphProto.initializer (5) -> TensorProto.dims (1), name (8)).  Data only:

    python tools/onnx_op_histogram.py       # -> tests/golden/ref_model_ops.json

`tests/test_model_shapes.py` checks what `tools/build_graphs.py` emits for nanoGPT / microgpt against these counts, through the operator
decomposition of the loader's handlers (atlas-onnx-tracer/src/node/handlers/*.rs)."""
import json
import os

MODELS = "/root/reference/atlas-onnx-tracer/models"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "ref_model_ops.json")


def varint(b, i):
    v, s = 0, 0
    while True:
        x = b[i]; i += 1
        v |= (x & 0x7F) << s
        if x < 0x80:
            return v, i
        s += 7


def fields(b):
    """(field number, wire type, value) of one message; value = int for varints / fixed, bytes for length-delimited"""
    i = 0
    while i < len(b):
        key, i = varint(b, i)
        f, wt = key >> 3, key & 7
        if wt == 0:
            v, i = varint(b, i)
        elif wt == 1:
            v, i = b[i:i + 8], i + 8
        elif wt == 2:
            n, i = varint(b, i)
            v, i = b[i:i + n], i + n
        elif wt == 5:
            v, i = b[i:i + 4], i + 4
        else:
            raise ValueError(f"wire type {wt}")
        yield f, wt, v


def packed_varints(b):
    out, i = [], 0
    while i < len(b):
        v, i = varint(b, i)
        out.append(v)
    return out


def model_info(path):
    data = open(path, "rb").read()
    graph = next(v for f, wt, v in fields(data) if f == 7 and wt == 2)
    ops, inits, n_nodes = {}, {}, 0
    for f, wt, v in fields(graph):
        if f == 1 and wt == 2:                         # NodeProto
            n_nodes += 1
            op = next((x.decode() for ff, w, x in fields(v) if ff == 4 and w == 2), "?")
            ops[op] = ops.get(op, 0) + 1
        elif f == 5 and wt == 2:                       # TensorProto initializer
            dims, name = [], ""
            for ff, w, x in fields(v):
                if ff == 1:
                    dims += packed_varints(x) if w == 2 else [x]
                elif ff == 8 and w == 2:
                    name = x.decode()
            inits[name] = dims
    return {"n_nodes": n_nodes, "ops": dict(sorted(ops.items())), "initializer_dims": inits}


def main():
    out = {"source": "atlas-onnx-tracer/models/*/network.onnx (operator types and initializer shapes; data only)", "generator": "tools/onnx_op_histogram.py", "models": {}}
    for name in sorted(os.listdir(MODELS)):
        p = os.path.join(MODELS, name, "network.onnx")
        if os.path.exists(p) and os.path.getsize(p) < (64 << 20):
            try:
                out["models"][name] = model_info(p)
            except Exception as e:      # a model this walker cannot read is named, not dropped
                out["models"][name] = {"error": str(e)}
    with open(OUT, "w") as f:
        json.dump(out, f, separators=(",", ":"))
        f.write("\n")
    for k in ("microgpt", "nanoGPT"):
        m = out["models"].get(k, {})
        print(k, m.get("n_nodes"), m.get("ops"))
        print("   ", {n: d for n, d in list(m.get("initializer_dims", {}).items())[:40]})


if __name__ == "__main__":
    main()
