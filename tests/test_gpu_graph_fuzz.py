"""GPU: randomly composed operator chains — shapes, operand orders and saturating values nobody picked by hand — through atlas_prove_graph
against the oracle composition (proof bytes, transcript state, the trace of every node) and through atlas_verify_graph (accepted with the
prover's state; one flipped byte rejected).  Seeded: a failing seed reproduces."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def random_chain(seed, steps=9, wide=False, sizes=(2, 4, 8)):
    rng = np.random.default_rng(seed)
    m, n = int(rng.choice(sizes)), int(rng.choice(sizes))
    nodes = [{"idx": 0, "op": "Input", "inputs": [], "dims": [m, n]}]
    x_in = rng.integers(-(1 << 11), 1 << 11, size=m * n).astype(np.int32)
    cur, dims = 0, [m, n]
    bits = {0: 11}                      # a bound on log2 |value| per node: products must stay inside the i64 accumulators (an overflowing
                                        # witness is not a valid one: prover and oracle wrap alike, the verifier rejects)

    def add(op, inputs, d, **kw):
        nd = {"idx": len(nodes), "op": op, "inputs": list(inputs), "dims": list(d)}
        nd.update(kw)
        nodes.append(nd)
        b = [bits[i] for i in inputs]
        sc = kw.get("scale", 0)
        bits[nd["idx"]] = min(31, max(0, {"Add": lambda: max(b) + 1, "Sub": lambda: max(b) + 1, "Mul": lambda: sum(b) - sc, "Square": lambda: 2 * b[0] - sc,
                                          "Cube": lambda: 3 * b[0] - 2 * sc, "Clamp": lambda: min(b[0], 9), "Einsum": lambda: sum(b) + 3 - sc,
                                          "Sum": lambda: b[0] + 3, "Concat": lambda: max(b)}.get(op, lambda: (b[0] if b else kw.get("_bits", 0)))()))
        return nd["idx"]

    def const(d, lim):
        i = add("Constant", [], d, data=rng.integers(-lim, lim, size=int(np.prod(d))).astype(np.int32))
        bits[i] = int(np.log2(lim))
        return i

    def same_shape_partner():
        prev = [nd["idx"] for nd in nodes if nd["dims"] == dims and nd["idx"] != cur and nd["op"] != "Constant"]
        if prev and rng.random() < 0.4:
            return int(rng.choice(prev))
        return const(dims, 1 << int(rng.integers(4, 20)))

    for _ in range(steps):
        r = rng.random()
        if wide and rng.random() < 0.45:
            r = 0.94 + 0.06 * rng.random()
        if r < 0.22:
            op = str(rng.choice(["Add", "Sub"]))
            p = same_shape_partner()
            cur = add(op, [cur, p] if rng.random() < 0.5 else [p, cur], dims)
        elif r < 0.34:
            p = same_shape_partner()
            if bits[cur] + bits[p] < 62:
                cur = add("Mul", [cur, p], dims, scale=int(rng.integers(3, 15)))
        elif r < 0.40:
            op = str(rng.choice(["Square", "Cube"]))
            if (3 if op == "Cube" else 2) * bits[cur] < 62:
                cur = add(op, [cur], dims, scale=int(rng.integers(4, 12)))
        elif r < 0.52:
            op = str(rng.choice(["ReLU", "Neg", "Clamp"]))
            cur = add(op, [cur], dims, **({"bound_log": 9} if op == "Clamp" else {}))
        elif r < 0.60 and len(dims) == 2 and dims[1] >= 2:       # (a contraction over one element is a sumcheck of zero rounds: not composed)
            p = int(rng.choice([2, 4, 8]))
            if bits[cur] + 13 >= 62:
                continue
            w = const([dims[1], p], 1 << 10)
            cur = add("Einsum", [cur, w], [dims[0], p], layout="mk,kn->mn", scale=int(rng.integers(6, 13)), shape=[dims[0], dims[1], p])
            dims = [dims[0], p]
        elif r < 0.68 and len(dims) == 2:
            dims = [dims[1], dims[0]]
            cur = add("MoveAxis", [cur], dims, source=0, destination=1)
        elif r < 0.75:
            tot = int(np.prod(dims))
            a = int(rng.choice([d for d in (1, 2, 4, 8, 16, 32, 64) if d <= tot and tot % d == 0]))
            dims = [a, tot // a] if a != tot else [tot]
            if len(dims) == 2 and dims[1] == 1:
                dims = [dims[0]]
            cur = add("Reshape", [cur], dims)
        elif r < 0.83:
            ax = int(rng.integers(0, len(dims)))
            if dims[ax] >= 2:
                half = dims[ax] // 2
                st = int(rng.choice([0, half]))
                nd_ = list(dims); nd_[ax] = half
                cur = add("Slice", [cur], nd_, axis=ax, start=st, end=st + half)
                dims = nd_
        elif r < 0.91:
            ax = int(rng.integers(0, len(dims)))
            if int(np.prod(dims)) <= 64:
                other = same_shape_partner()
                nd_ = list(dims); nd_[ax] = 2 * dims[ax]
                cur = add("Concat", [cur, other] if rng.random() < 0.5 else [other, cur], nd_, axis=ax)
                dims = nd_
        elif r < (0.94 if wide else 2.0) and len(dims) == 2 and dims[1] >= 2:
            s = add("Sum", [cur], [dims[0], 1], axes=[1])
            b = add("Broadcast", [s], dims)
            cur = add("Sub", [cur, b], dims)
        elif wide and r < 0.95:
            cur = add("ScalarConstDiv", [cur], dims, divisor=int(rng.integers(2, 100)))
        elif wide and r < 0.96:
            d_ = add("Constant", [], dims, data=rng.integers(1, 1 << int(rng.integers(1, 12)), size=int(np.prod(dims))).astype(np.int32))
            bits[d_] = 12
            cur = add("Div", [cur, d_], dims)
        elif wide and r < 0.97 and len(dims) == 2 and dims[1] >= 2 and 2 * bits[cur] + 3 < 62:
            S = int(rng.integers(4, 12))
            ms = add("MeanOfSquares", [cur], [dims[0], 1], axes=[1], scale=S, count=dims[1])
            bits[ms] = min(31, max(0, 2 * bits[cur] - S))
            eps = add("Constant", [], [dims[0], 1], data=np.full(dims[0], int(rng.integers(1, 5)), dtype=np.int32)); bits[eps] = 3
            v_ = add("Add", [ms, eps], [dims[0], 1])
            rs = add("Rsqrt", [v_], [dims[0], 1], scale=S); bits[rs] = 3 * S // 2 + 1
            b = add("Broadcast", [rs], dims)
            if bits[cur] + bits[b] < 62:
                cur = add("Mul", [cur, b], dims, scale=S)
            else:
                cur = add("Add", [cur, b], dims)
        elif wide and r < 0.98:
            op = str(rng.choice(["Tanh", "Erf", "Sigmoid", "Sin", "Cos"]))
            cur = add(op, [cur], dims, scale=14); bits[cur] = 15
        elif wide and r < 0.99 and len(dims) == 2 and dims[0] >= 2 and dims[1] >= 2:
            cur = add("SoftmaxLastAxis", [cur], dims, scale=14); bits[cur] = 15
        elif wide:
            mask = add("Constant", [], dims, data=rng.integers(0, 2, size=int(np.prod(dims))).astype(np.int32)); bits[mask] = 1
            other = same_shape_partner()
            cur = add("Iff", [mask, cur, other], dims); bits[cur] = max(bits[nodes[cur]["inputs"][1]], bits[other])
    if nodes[cur]["op"] in ("Input", "Constant"):
        cur = add("ReLU", [cur], dims)
    return nodes, [cur], [x_in]


@pytest.mark.parametrize("seed", list(range(14)))
def test_random_operator_chain(atlas, seed):
    from oracle import graph as OG, orc
    from jolt_atlas_amd import graph as GG
    nodes, outputs, inputs = random_chain(1000 + seed)
    nv = 4 + max(int(np.log2(max(int(np.prod(nd["dims"])), 1))) for nd in nodes)
    tau = orc.random_fr(1, 0x51250002)[0]
    srs = atlas.SRS.generate(tau, 1 << nv)
    P = OG.Prover(nodes, outputs, orc.srs_powers(tau, 1 << nv))
    want = P.prove(inputs)
    G = GG.Graph(nodes, outputs)
    got, state, _ = G.prove(srs, inputs)
    for nd in nodes:
        assert np.array_equal(G.node_output(nd["idx"]), P.trace[nd["idx"]]), f"trace of node {nd['idx']} ({nd['op']}), seed {seed}"
    assert state == P.t.state() and got == want, [nd["op"] for nd in nodes]
    vk = atlas.HyperKZG.vk_from_trapdoor(tau, srs.download(0, 1)[0])
    V = GG.Graph(nodes, outputs)
    out = G.node_output(outputs[0])
    ok, vstate = V.verify(vk, inputs, out, got)
    assert ok and vstate == state
    off = int(np.random.default_rng(seed).integers(8, len(got)))
    bad = bytearray(got); bad[off] ^= 1
    try:
        assert not V.verify(vk, inputs, out, bytes(bad))[0], off
    except atlas.AtlasError:
        pass
    G.free(); V.free(); srs.free()


def _run(atlas, nodes, outputs, inputs, seed):
    from oracle import graph as OG, orc
    from jolt_atlas_amd import graph as GG
    nv = 4 + max(int(np.log2(max(int(np.prod(nd["dims"])), 1))) for nd in nodes)
    tau = orc.random_fr(1, 0x51250004)[0]
    srs = atlas.SRS.generate(tau, 1 << nv)
    P = OG.Prover(nodes, outputs, orc.srs_powers(tau, 1 << nv))
    want = P.prove(inputs)
    G = GG.Graph(nodes, outputs)
    got, state, _ = G.prove(srs, inputs)
    for nd in nodes:
        assert np.array_equal(G.node_output(nd["idx"]), P.trace[nd["idx"]]), f"trace of node {nd['idx']} ({nd['op']}), seed {seed}"
    assert state == P.t.state() and got == want, [(nd["op"], nd["dims"]) for nd in nodes]
    vk = atlas.HyperKZG.vk_from_trapdoor(tau, srs.download(0, 1)[0])
    V = GG.Graph(nodes, outputs)
    ok, vstate = V.verify(vk, inputs, G.node_output(outputs[0]), got)
    assert ok and vstate == state, [(nd["op"], nd["dims"]) for nd in nodes]
    G.free(); V.free(); srs.free()


@pytest.mark.parametrize("seed", list(range(12)))
def test_random_operator_chain_tiny_shapes(atlas, seed):
    """axes of one and two elements: scalar nodes (no lookups, zero-round sumchecks, clear-text checks in the verifier) between vector ones"""
    nodes, outputs, inputs = random_chain(9000 + seed, steps=8, sizes=(1, 2))
    # (round 4: the one-element corner cases are composed — a scalar-output Einsum, Div, and the lookup operators over one cycle:
    # tests/test_gpu_one_element.py — so nothing of this draw is skipped any more)
    _run(atlas, nodes, outputs, inputs, seed)


@pytest.mark.parametrize("seed", list(range(10)))
def test_random_operator_chain_wide(atlas, seed):
    """the same with the lookup-heavy operators in the draw: ScalarConstDiv, Div, MeanOfSquares + Rsqrt, Tanh / Erf / Sigmoid / Sin / Cos, SoftmaxLastAxis, Iff"""
    from oracle import graph as OG, orc
    from jolt_atlas_amd import graph as GG
    nodes, outputs, inputs = random_chain(5000 + seed, steps=11, wide=True)
    nv = 4 + max(int(np.log2(max(int(np.prod(nd["dims"])), 1))) for nd in nodes)
    tau = orc.random_fr(1, 0x51250003)[0]
    srs = atlas.SRS.generate(tau, 1 << nv)
    P = OG.Prover(nodes, outputs, orc.srs_powers(tau, 1 << nv))
    want = P.prove(inputs)
    G = GG.Graph(nodes, outputs)
    got, state, _ = G.prove(srs, inputs)
    for nd in nodes:
        assert np.array_equal(G.node_output(nd["idx"]), P.trace[nd["idx"]]), f"trace of node {nd['idx']} ({nd['op']}), seed {seed}"
    assert state == P.t.state() and got == want, [nd["op"] for nd in nodes]
    vk = atlas.HyperKZG.vk_from_trapdoor(tau, srs.download(0, 1)[0])
    V = GG.Graph(nodes, outputs)
    ok, vstate = V.verify(vk, inputs, G.node_output(outputs[0]), got)
    assert ok and vstate == state, [nd["op"] for nd in nodes]
    G.free(); V.free(); srs.free()
