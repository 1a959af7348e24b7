"""How a doc-test record of the reference's tensor ops (tests/golden/ref_tensor_ops.json, made by tools/extract_ref_doctests.py)
maps onto the operators `Model::trace` executes here.

plan(case) returns one of
    ("graph", nodes, inputs, out_idx, witness_key)   a graph description in the tracer's vocabulary with the record's RAW dims
                                                     (oracle/graph.py:execute takes any dims); `pad_plan` turns it into the
                                                     power-of-two description the device takes, or says why it cannot
    ("formula", fn)                                  a scalar formula of the oracle (the activation / trig tables' generator) at
                                                     the record's multiplier
    ("unmapped", reason)                             no operator of ONNXProof executes this function (or not with these arguments)

The mapping follows which tensor op each operator's `Op::f` calls (atlas-onnx-tracer/src/ops/*.rs)."""
import numpy as np

# the prover's einsum layouts (jolt-atlas-core/src/onnx_proof/ops/einsum/{mk_kn_mn,k_nk_n,bmk_rhs_mbn,mbk_rhs_bmn,rbmk_rbnk_bmn}.rs)
LAYOUTS = ["mk,kn->mn", "k,nk->n", "bmk,bkn->mbn", "bmk,kbn->mbn", "mbk,bnk->bmn", "mbk,nbk->bmn"]

NOT_AN_OPERATOR = "no ONNXProof operator executes this tensor function"


def tensor(t):
    return np.array(t["data"], dtype=np.int64).reshape(t["dims"])


def canon_equation(eq):
    """rename the letters of an einsum equation in order of first appearance"""
    m, out = {}, ""
    for ch in eq:
        if ch.isalpha():
            m.setdefault(ch, chr(ord("a") + len(m)))
            out += m[ch]
        else:
            out += ch
    return out


def match_layout(eq, shapes):
    """the prover layout an equation is a renaming of, with its `shape` parameter; None if it is outside the set"""
    for lay in LAYOUTS:
        if canon_equation(lay) != canon_equation(eq):
            continue
        ins = lay.split("->")[0].split(",")
        size = {}
        for names, shp in zip(ins, shapes):
            if len(names) != len(shp):
                return None
            for ch, d in zip(names, shp):
                size[ch] = d
        order = {"mk,kn->mn": "mkn", "k,nk->n": "kn"}.get(lay, "bmkn")
        return lay, [size[c] for c in order]
    return None


def _inputs_and_nodes(tensors):
    nodes, inputs = [], []
    for i, t in enumerate(tensors):
        nodes.append({"idx": i, "op": "Input", "inputs": [], "dims": list(t["dims"])})
        inputs.append(np.array(t["data"], dtype=np.int32))
    return nodes, inputs


def _binary_chain(case, op, extra=None):
    ts = case["args"][0]["tensors"]
    if len(ts) != 2:
        return ("unmapped", f"{op} of {len(ts)} operands in one call (the operator is binary)")
    nodes, inputs = _inputs_and_nodes(ts)
    out_dims = list(case["expected"]["dims"])
    ids = [0, 1]
    for k, t in enumerate(ts):
        if list(t["dims"]) != out_dims:                      # HandlerBuilder::with_broadcast: a Broadcast node in front
            src = k
            if len(t["dims"]) != len(out_dims):
                if int(np.prod(t["dims"])) != 1:
                    return ("unmapped", "broadcast across ranks of a tensor with more than one element")
                nodes.append({"idx": len(nodes), "op": "Reshape", "inputs": [k], "dims": [1] * len(out_dims)})
                src = len(nodes) - 1
            nodes.append({"idx": len(nodes), "op": "Broadcast", "inputs": [src], "dims": out_dims})
            ids[k] = len(nodes) - 1
    nd = {"idx": len(nodes), "op": op, "inputs": ids, "dims": out_dims}
    nd.update(extra or {})
    nodes.append(nd)
    return ("graph", nodes, inputs, nd["idx"], None)


def plan(case):
    fn, args, exp = case["fn"], case["args"], case["expected"]
    if fn in ("add", "sub"):
        return _binary_chain(case, "Add" if fn == "add" else "Sub")
    if fn == "mult":
        return ("unmapped", "tensor::ops::mult is Mul with scale 0, the raw building-block product (ops/mul.rs:13-15); the loader always emits the "
                            "fused form (handlers/arith.rs:43-48) and only that one is modelled")
    if fn == "iff":
        nodes, inputs = _inputs_and_nodes([a["tensor"] for a in args])
        nodes.append({"idx": 3, "op": "Iff", "inputs": [0, 1, 2], "dims": list(exp["dims"])})
        return ("graph", nodes, inputs, 3, None)
    if fn == "and":
        nodes, inputs = _inputs_and_nodes([a["tensor"] for a in args])
        nodes.append({"idx": 2, "op": "And", "inputs": [0, 1], "dims": list(exp["dims"])})
        return ("graph", nodes, inputs, 2, None)
    if fn == "neg":
        nodes, inputs = _inputs_and_nodes([args[0]["tensor"]])
        nodes.append({"idx": 1, "op": "Neg", "inputs": [0], "dims": list(exp["dims"])})
        return ("graph", nodes, inputs, 1, None)
    if fn == "einsum":
        ts = args[1]["tensors"]
        if len(ts) != 2:
            return ("unmapped", "einsum of three operands (the prover's layouts are binary)")
        eq = args[0]["str"]
        m = match_layout(eq, [t["dims"] for t in ts])
        if m is None:                                            # the same contraction with the operands in the other order
            lhs, rhs = eq.split("->")
            a, b = lhs.split(",")
            m = match_layout(f"{b},{a}->{rhs}", [ts[1]["dims"], ts[0]["dims"]])
            if m is not None:
                ts = [ts[1], ts[0]]
        if m is None:
            return ("unmapped", f"equation {eq} is outside the prover's layouts {LAYOUTS}")
        lay, shape = m
        nodes, inputs = _inputs_and_nodes(ts)
        # the fused operator rescales by 2^scale >= 2; scale 1 over a doubled left operand is the record's plain contraction
        inputs[0] = inputs[0] * 2
        nodes.append({"idx": 2, "op": "Einsum", "inputs": [0, 1], "dims": list(exp["dims"]), "layout": lay, "scale": 1, "shape": shape})
        return ("graph", nodes, inputs, 2, None)
    if fn == "sum_axes":
        nodes, inputs = _inputs_and_nodes([args[0]["tensor"]])
        nodes.append({"idx": 1, "op": "Sum", "inputs": [0], "dims": list(exp["dims"]), "axes": list(args[1]["list"])})
        return ("graph", nodes, inputs, 1, None)
    if fn == "gather":
        if args[2]["num"] != 0:
            return ("unmapped", "gather along axis 1 (the Gather operators index rows of a dictionary: axis 0, ops/gather/mod.rs)")
        d, ix = args[0]["tensor"], args[1]["tensor"]
        nodes, inputs = _inputs_and_nodes([d, ix])
        nodes.append({"idx": 2, "op": "GatherSmall", "inputs": [0, 1], "dims": list(exp["dims"]), "axis": 0, "dict_len": d["dims"][0]})
        return ("graph", nodes, inputs, 2, None)
    if fn == "concat":
        ts = args[0]["tensors"]
        nodes, inputs = _inputs_and_nodes(ts)
        nodes.append({"idx": len(ts), "op": "Concat", "inputs": list(range(len(ts))), "dims": list(exp["dims"]), "axis": args[1]["num"]})
        return ("graph", nodes, inputs, len(ts), None)
    if fn == "slice":
        nodes, inputs = _inputs_and_nodes([args[0]["tensor"]])
        nodes.append({"idx": 1, "op": "Slice", "inputs": [0], "dims": list(exp["dims"]), "axis": args[1]["num"], "start": args[2]["num"],
                      "end": args[3]["num"]})
        return ("graph", nodes, inputs, 1, None)
    if fn == "sra":
        sh = set(args[1]["tensor"]["data"])
        if len(sh) != 1:
            return ("unmapped", "per-element shift amounts")
        # an arithmetic right shift by S is the fused rescale of x * 1 by 2^S (floor_rebase_i64, ops/mod.rs:224-232)
        x = args[0]["tensor"]
        nodes, inputs = _inputs_and_nodes([x, {"data": [1] * len(x["data"]), "dims": x["dims"]}])
        nodes.append({"idx": 2, "op": "Mul", "inputs": [0, 1], "dims": list(exp["dims"]), "scale": sh.pop()})
        return ("graph", nodes, inputs, 2, None)
    if fn == "div":
        nodes, inputs = _inputs_and_nodes([args[0]["tensor"], args[1]["tensor"]])
        nodes.append({"idx": 2, "op": "Div", "inputs": [0, 1], "dims": list(exp["dims"])})
        return ("graph", nodes, inputs, 2, None)
    if fn in ("const_div", "const_rem"):
        k = args[1]["num"]
        if int(k) != k:
            return ("unmapped", "non-integer divisor")
        nodes, inputs = _inputs_and_nodes([args[0]["tensor"]])
        nodes.append({"idx": 1, "op": "ScalarConstDiv", "inputs": [0], "dims": list(exp["dims"]), "divisor": int(k)})
        return ("graph", nodes, inputs, 1, "rem" if fn == "const_rem" else None)
    if fn == "mean_of_squares_axes":
        x = args[0]["tensor"]
        axes = list(args[1]["list"])
        if axes != [len(x["dims"]) - 1]:
            return ("unmapped", "MeanOfSquares reduces the last axis")
        nodes, inputs = _inputs_and_nodes([x])
        # scale 0: D = 2^0 * count, the record's plain mean (ops/mean_of_squares.rs)
        nodes.append({"idx": 1, "op": "MeanOfSquares", "inputs": [0], "dims": list(exp["dims"]), "axes": axes, "scale": 0, "count": x["dims"][-1]})
        return ("graph", nodes, inputs, 1, None)
    if fn == "Tensor::move_axis":                            # the MoveAxis operator (ops/move_axis.rs -> Tensor::move_axis)
        nodes, inputs = _inputs_and_nodes([args[0]["tensor"]])
        nodes.append({"idx": 1, "op": "MoveAxis", "inputs": [0], "dims": list(exp["dims"]), "source": args[1]["num"], "destination": args[2]["num"]})
        return ("graph", nodes, inputs, 1, None)
    if fn == "Tensor::expand":                               # the Broadcast operator (ops/broadcast.rs -> Tensor::expand)
        nodes, inputs = _inputs_and_nodes([args[0]["tensor"]])
        nodes.append({"idx": 1, "op": "Broadcast", "inputs": [0], "dims": list(args[1]["list"])})
        return ("graph", nodes, inputs, 1, None)
    if fn == "max_axes":
        return ("formula", "softmax_max")
    if fn in ("tanh", "erffunc", "sigmoid", "sin", "cos"):
        return ("formula", {"tanh": "Tanh", "erffunc": "Erf", "sigmoid": "Sigmoid", "sin": "Sin", "cos": "Cos"}[fn])
    if fn == "leakyrelu":
        if args[1]["num"] != 0:
            return ("unmapped", "slope 0.1: ReLU is leakyrelu with slope 0 (ops/relu.rs:9)")
    return ("unmapped", NOT_AN_OPERATOR)


def next_pow2(d):
    return 1 << max(int(d) - 1, 0).bit_length()


def pad_plan(nodes, inputs, out_idx):
    """The same graph with every dimension padded to a power of two the way the loader pads (zeros; model/load.rs:505-559), or a reason why
    zero padding is not neutral for it.  Returns (nodes, inputs, crop) — crop = the raw dims of the output inside the padded ones."""
    by = {nd["idx"]: nd for nd in nodes}
    out, pin = [], []
    it = iter(inputs)
    for nd in nodes:
        q = dict(nd)
        q["dims"] = [next_pow2(d) for d in nd["dims"]]
        op = nd["op"]
        if op == "Input":
            x = np.asarray(next(it)).reshape(nd["dims"])
            fill = 0
            users = [u for u in nodes if nd["idx"] in u["inputs"]]
            if any(u["op"] == "Div" and u["inputs"][1] == nd["idx"] for u in users):
                fill = 1                                             # a padded divisor of 0 has no quotient
            p = np.full(q["dims"], fill, dtype=np.int32)
            p[tuple(slice(0, d) for d in nd["dims"])] = x
            pin.append(p.reshape(-1))
        elif op == "MoveAxis" and any(next_pow2(d) != d for d in nd["dims"]):
            return None, "MoveAxis of a shape with a dimension that is not a power of two: the padded elements would move with the axis"
        elif op == "Concat":
            ax = nd["axis"]
            if any(next_pow2(by[j]["dims"][ax]) != by[j]["dims"][ax] for j in nd["inputs"]) or next_pow2(nd["dims"][ax]) != nd["dims"][ax]:
                return None, "Concat along an axis that is not a power of two: padding would move the second operand"
        elif op == "Slice":
            if next_pow2(nd["end"] - nd["start"]) != nd["end"] - nd["start"]:
                return None, "Slice of a length that is not a power of two"
        elif op == "Einsum":
            q["shape"] = [next_pow2(d) for d in nd["shape"]]
        elif op == "GatherSmall":
            q["dict_len"] = next_pow2(nd["dict_len"])
        out.append(q)
    return (out, pin, list(by[out_idx]["dims"])), None


def crop(flat, padded_dims, raw_dims):
    a = np.asarray(flat).reshape(padded_dims)
    return a[tuple(slice(0, d) for d in raw_dims)].reshape(-1)
