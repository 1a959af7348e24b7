#!/usr/bin/env python3
"""Pull the literal test vectors the reference HOLDS for the layer under `Model::trace` out of its doc tests:

    /root/reference/atlas-onnx-tracer/src/tensor/ops.rs     (the `/// ``` ... /// ```` example of every tensor op)

Only DATA leaves the reference: per `assert_eq!(result, expected)` one record with the function's name, the line of the
example, the argument tensors / scalars / equation string and the expected tensor.  No reference source text is written.
Run in the build container (the reference is not on the GPU box):

    python tools/extract_ref_doctests.py            # -> tests/golden/ref_tensor_ops.json

`tests/test_ref_tensor_ops.py` replays the records through `oracle/graph.py:execute` (CPU) and `atlas_graph_trace` (GPU)."""
import json
import os
import re
import sys

REF = "/root/reference/atlas-onnx-tracer/src/tensor/ops.rs"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "ref_tensor_ops.json")


def doc_blocks(lines):
    """(first line number, function name, [statement text]) per fenced example; a fence is a doc line that is only ```"""
    def doc_text(l):
        s = l.strip()
        if not s.startswith("///"):
            return None
        return s[3:].strip()

    def is_fence(l):
        t = doc_text(l)
        return t is not None and t.lstrip("/ ").startswith("```")
    i, out = 0, []
    while i < len(lines):
        if is_fence(lines[i]):
            j, body = i + 1, []
            while j < len(lines) and not is_fence(lines[j]):
                t = doc_text(lines[j])
                if t is None:
                    break
                body.append(t)
                j += 1
            k = j + 1
            while k < len(lines) and not re.match(r"\s*pub (?:\(crate\) )?fn (\w+)", lines[k]):
                k += 1
            name = re.match(r"\s*pub (?:\(crate\) )?fn (\w+)", lines[k]).group(1) if k < len(lines) else "?"
            out.append((i + 1, name, body))
            i = j + 1
        else:
            i += 1
    return out


def split_top(s, sep):
    """split on `sep` outside brackets and string literals"""
    parts, depth, cur, in_str = [], 0, "", False
    for ch in s:
        if ch == '"':
            in_str = not in_str
        if not in_str:
            if ch in "([{":
                depth += 1
            elif ch in ")]}":
                depth -= 1
            elif ch == sep and depth == 0:
                parts.append(cur.strip())
                cur = ""
                continue
        cur += ch
    if cur.strip():
        parts.append(cur.strip())
    return parts


NUM = r"-?\d[\d_]*(?:\.\d+)?(?:e-?\d+)?"


def parse_number(tok):
    t = tok.strip().lstrip("&").strip()
    t = re.sub(r"_?(i32|i64|usize|u32|u64|u128|f64|f32|i128)$", "", t).replace("_", "")
    if re.fullmatch(r"-?\d+", t):
        return int(t)
    if re.fullmatch(r"-?\d+\.\d*(e-?\d+)?|-?\d+e-?\d+", t):
        return float(t)
    return None


def parse_tensor(rhs):
    m = re.search(r"Tensor(?:::<\s*(\w+)\s*>)?::new\(\s*Some\(\s*&\s*\[(.*?)\]\s*\)\s*,\s*&\s*\[(.*?)\]\s*,?\s*\)", rhs, re.S)
    if not m:
        return None
    data = [parse_number(x) for x in split_top(m.group(2), ",")]
    dims = [parse_number(x) for x in split_top(m.group(3), ",")]
    if any(v is None for v in data + dims):
        return None
    ty = m.group(1)
    if ty is None:
        t0 = re.search(r"\d(_?)(i32|usize|i64|f64)", m.group(2))
        ty = t0.group(2) if t0 else "i32"
    return {"type": ty, "data": data, "dims": dims}


def parse_arg(tok, env):
    t = tok.strip()
    t = re.sub(r"\.clone\(\)$", "", t)
    if t.startswith('"'):
        return {"str": t.strip('"')}
    if t.startswith("&[") or t.startswith("["):
        inner = t[t.index("[") + 1:t.rindex("]")]
        items = [parse_arg(x, env) for x in split_top(inner, ",")]
        if any(x is None for x in items):
            return None
        if all("tensor" in x for x in items):
            return {"tensors": [x["tensor"] for x in items]}
        if all("num" in x for x in items):
            return {"list": [x["num"] for x in items]}
        return None
    name = re.sub(r"\.clone\(\)$", "", t.lstrip("&").strip())
    if name in env:
        v = env[name]
        return {"tensor": v} if isinstance(v, dict) else {"num": v}
    n = parse_number(t)
    if n is not None:
        return {"num": n}
    return None


def parse_block(line, fn, body):
    text = " ".join(re.sub(r"(^|\s)//.*$", "", l) for l in body)       # examples carry // comments
    env, cases, skipped = {}, [], []
    call = None
    for st in split_top(text, ";"):
        st = st.strip()
        if not st or st.startswith("use "):
            continue
        m = re.match(r"let\s+(?:mut\s+)?(\w+)\s*(?::[^=]+)?=\s*(.*)$", st, re.S)
        if m:
            name, rhs = m.group(1), m.group(2).strip()
            if "Tensor" in rhs and "new(" in rhs:
                t = parse_tensor(rhs)
                if t is None:
                    skipped.append(f"tensor literal not parsed: {name}")
                    env.pop(name, None)
                else:
                    env[name] = t
                continue
            n = parse_number(rhs)
            if n is not None:
                env[name] = n
                continue
            c = re.match(r"(\w+)\((.*)\)\s*$", re.sub(r"\.unwrap\(\)\s*$", "", rhs), re.S)
            if c and name == "result":
                args = [parse_arg(a, env) for a in split_top(c.group(2), ",")]
                call = None if any(a is None for a in args) else {"fn": c.group(1), "args": args}
                if call is None:
                    skipped.append(f"call arguments not parsed: {rhs[:60]}")
                continue
            env.pop(name, None)
            continue
        if st.startswith("assert_eq!"):
            inner = st[st.index("(") + 1:st.rindex(")")]
            a, b = [x.strip() for x in split_top(inner, ",")][:2]
            if a == "result" and call is not None and isinstance(env.get(b), dict):
                cases.append({"fn": call["fn"], "line": line, "args": call["args"], "expected": env[b]})
            else:
                skipped.append(f"assert not in the result / expected form: {st[:60]}")
    return cases, skipped


REF_METHODS = "/root/reference/atlas-onnx-tracer/src/tensor/mod.rs"


def method_cases():
    """Tensor::move_axis / Tensor::expand (tensor/mod.rs: what the MoveAxis and Broadcast operators execute): the method-call examples
    `let b = a.move_axis(s, d).unwrap(); assert_eq!(b, expected)` and `assert_eq!(a.expand(&[..]).unwrap(), expected)`"""
    out = []
    for line, fn, body in doc_blocks(open(REF_METHODS).read().split("\n")):
        if fn not in ("move_axis", "expand"):
            continue
        text = " ".join(re.sub(r"(^|\s)//.*$", "", l) for l in body)
        env, res = {}, {}
        for st in split_top(text, ";"):
            st = st.strip()
            m = re.match(r"let\s+(?:mut\s+)?(\w+)\s*(?::[^=]+)?=\s*(.*)$", st, re.S)
            if m and "Tensor" in m.group(2) and "new(" in m.group(2):
                t = parse_tensor(m.group(2))
                if t is not None:
                    env[m.group(1)] = t
                continue
            call = re.search(r"(\w+)\.(move_axis|expand)\((.*?)\)\.unwrap\(\)", st)
            if m and call and call.group(1) in env:
                res[m.group(1)] = (call.group(2), env[call.group(1)], [parse_arg(a, env) for a in split_top(call.group(3), ",")])
                continue
            if st.startswith("assert_eq!"):
                inner = st[st.index("(") + 1:st.rindex(")")]
                a, b = [x.strip() for x in split_top(inner, ",")][:2]
                got = res.get(a)
                if got is None and call and call.group(1) in env:
                    got = (call.group(2), env[call.group(1)], [parse_arg(x, env) for x in split_top(call.group(3), ",")])
                if got is not None and isinstance(env.get(b), dict) and all(x is not None for x in got[2]):
                    out.append({"fn": "Tensor::" + got[0], "line": line, "args": [{"tensor": got[1]}] + got[2], "expected": env[b], "documents": fn,
                                "file": "tensor/mod.rs"})
    return out


def main():
    lines = open(REF).read().split("\n")
    cases, skipped, n_blocks = [], [], 0
    for line, fn, body in doc_blocks(lines):
        n_blocks += 1
        c, s = parse_block(line, fn, body)
        for x in c:
            x["documents"] = fn
        cases += c
        skipped += [{"line": line, "fn": fn, "why": w} for w in s]
    cases += method_cases()
    out = {"source": "atlas-onnx-tracer/src/tensor/ops.rs + Tensor::move_axis / Tensor::expand of tensor/mod.rs (doc tests; data only)",
           "generator": "tools/extract_ref_doctests.py", "n_blocks": n_blocks, "n_cases": len(cases), "cases": cases, "unparsed": skipped}
    with open(OUT, "w") as f:
        json.dump(out, f, separators=(",", ":"))
        f.write("\n")
    print(f"{n_blocks} examples, {len(cases)} assert_eq! records, {len(skipped)} unparsed -> {OUT}")
    for s in skipped:
        print("  unparsed:", s)


if __name__ == "__main__":
    sys.exit(main())
