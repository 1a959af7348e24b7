"""One-element corner cases that the reference's flows cover by construction (`-m gpu`): a ScalarConstDiv and a Div of ONE element — a
committed polynomial with one coefficient, i.e. a zero-round member of the opening reduction (`opening_proof.rs:447-532`; Div:
`ops/div.rs:93-160`, `if node.is_scalar()`: the quotient only, no range check) —, an Einsum with a scalar output (fused rescale,
`is_scalar`: the accumulation opens in the clear, `fused_rebase.rs:224-310`), and the lookup operators proper over ONE element (ReLU, Clamp,
Tanh / Erf / Sigmoid, Rsqrt, Sin / Cos: no is_scalar branch in the reference, so their generic flows run over one cycle — read-raf instances
without cycle variables, `ps_shout/mod.rs:419-446`, one-hot checks over one cycle, one-hot commitments of K x 1 coefficients).  Each graph: the device's proof against the in-repo oracle's,
then ONNXProof::verify of the device's proof, and a wrong output rejected."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _wrap(mid, x0):
    """Input[2,2] -> Sum -> Sum -> one element s; mid(s) -> one-element node(s); Broadcast -> [1,2]; Add of a constant"""
    nodes = [{"idx": 0, "op": "Input", "inputs": [], "dims": [2, 2]},
             {"idx": 1, "op": "Sum", "inputs": [0], "dims": [2, 1], "axes": [1]},
             {"idx": 2, "op": "Sum", "inputs": [1], "dims": [1, 1], "axes": [0]}]
    nodes += mid(2, 3)
    last = nodes[-1]["idx"]
    nodes += [{"idx": last + 1, "op": "Broadcast", "inputs": [last], "dims": [1, 2]},
              {"idx": last + 2, "op": "Constant", "inputs": [], "dims": [1, 2], "data": np.array([3, -4], dtype=np.int32)},
              {"idx": last + 3, "op": "Add", "inputs": [last + 1, last + 2], "dims": [1, 2]}]
    return nodes, [last + 3], [np.asarray(x0, dtype=np.int32)]


def scd_graph():
    return _wrap(lambda s, i: [{"idx": i, "op": "ScalarConstDiv", "inputs": [s], "dims": [1, 1], "divisor": 7}], [11, -25, 300, 41])


def scd_negative_graph():
    return _wrap(lambda s, i: [{"idx": i, "op": "ScalarConstDiv", "inputs": [s], "dims": [1, 1], "divisor": 5}], [-11, -25, -300, -41])


def div_graph():
    return _wrap(lambda s, i: [{"idx": i, "op": "Constant", "inputs": [], "dims": [1, 1], "data": np.array([70000], dtype=np.int32)},
                               {"idx": i + 1, "op": "Div", "inputs": [i, s], "dims": [1, 1]}], [5000, 6000, 7000, 8000])


def einsum_scalar_graph():
    return ([{"idx": 0, "op": "Input", "inputs": [], "dims": [1, 4]},
             {"idx": 1, "op": "Constant", "inputs": [], "dims": [4, 1], "data": np.array([3, -2, 5, 7], dtype=np.int32)},
             {"idx": 2, "op": "Einsum", "inputs": [0, 1], "dims": [1, 1], "layout": "mk,kn->mn", "scale": 2, "shape": [1, 4, 1]},
             {"idx": 3, "op": "Broadcast", "inputs": [2], "dims": [1, 2]},
             {"idx": 4, "op": "Constant", "inputs": [], "dims": [1, 2], "data": np.array([1, 2], dtype=np.int32)},
             {"idx": 5, "op": "Add", "inputs": [3, 4], "dims": [1, 2]}], [5], [np.array([10, 20, -30, 40], dtype=np.int32)])


def relu_graph():
    return _wrap(lambda s, i: [{"idx": i, "op": "ReLU", "inputs": [s], "dims": [1, 1]}], [11, -25, 300, 41])


def relu_negative_graph():
    return _wrap(lambda s, i: [{"idx": i, "op": "ReLU", "inputs": [s], "dims": [1, 1]}], [-11, -25, -300, -41])


def clamp_graph():
    return _wrap(lambda s, i: [{"idx": i, "op": "Clamp", "inputs": [s], "dims": [1, 1], "bound_log": 9}], [1100, 2500, 300, 41])


def _act(op, x0):
    def g():
        return _wrap(lambda s, i: [{"idx": i, "op": op, "inputs": [s], "dims": [1, 1], "scale": 14}], x0)
    g.__name__ = op.lower() + "_graph"
    return g


tanh_graph, erf_graph, sigmoid_graph = _act("Tanh", [1100, 2500, 300, 41]), _act("Erf", [-1100, -2500, 300, 41]), _act("Sigmoid", [11000, 25000, 30000, 41])
rsqrt_graph = _act("Rsqrt", [5000, 6000, 7000, 8000])
sin_graph, cos_graph = _act("Sin", [500000, 600000, 700000, 8000]), _act("Cos", [-5000, -6000, -7000, -8000])

def _gather(op, V, idx, seed=5):
    """ONE index into a dictionary of V one-element words (ops/gather/{small,large}.rs have no is_scalar branch: the execution sumcheck runs over
    the log V address variables, the one-hot checks over one cycle)"""
    def g():
        rng = np.random.default_rng(seed)
        return ([{"idx": 0, "op": "Input", "inputs": [], "dims": [1]},
                 {"idx": 1, "op": "Constant", "inputs": [], "dims": [V, 1], "data": rng.integers(-(1 << 14), 1 << 14, size=V).astype(np.int32)},
                 {"idx": 2, "op": op, "inputs": [1, 0], "dims": [1, 1], "axis": 0, "dict_len": V},
                 {"idx": 3, "op": "Broadcast", "inputs": [2], "dims": [1, 2]},
                 {"idx": 4, "op": "Constant", "inputs": [], "dims": [1, 2], "data": np.array([1, 2], dtype=np.int32)},
                 {"idx": 5, "op": "Add", "inputs": [3, 4], "dims": [1, 2]}], [5], [np.array([idx], dtype=np.int32)])
    g.__name__ = "%s_%d_of_%d_graph" % (op.lower(), idx, V)
    return g


def _gather_row(op, V, word, idx, seed=11):
    """ONE index, a word of several elements (a single token's embedding row): no cycle variables, log word output variables"""
    def g():
        rng = np.random.default_rng(seed)
        return ([{"idx": 0, "op": "Input", "inputs": [], "dims": [1]},
                 {"idx": 1, "op": "Constant", "inputs": [], "dims": [V, word], "data": rng.integers(-(1 << 14), 1 << 14, size=V * word).astype(np.int32)},
                 {"idx": 2, "op": op, "inputs": [1, 0], "dims": [1, word], "axis": 0, "dict_len": V},
                 {"idx": 3, "op": "Constant", "inputs": [], "dims": [1, word], "data": rng.integers(-9, 9, size=word).astype(np.int32)},
                 {"idx": 4, "op": "Add", "inputs": [2, 3], "dims": [1, word]}], [4], [np.array([idx], dtype=np.int32)])
    g.__name__ = "%s_row_%d_of_%dx%d_graph" % (op.lower(), idx, V, word)
    return g


GATHERS = [_gather_row("GatherSmall", 16, 4, 9), _gather_row("GatherLarge", 64, 8, 63),
           _gather("GatherSmall", 4, 2), _gather("GatherSmall", 16, 0, 6), _gather("GatherSmall", 16, 15, 7),
           _gather("GatherLarge", 64, 37), _gather("GatherLarge", 256, 255, 8), _gather("GatherLarge", 64, 0, 9)]

def _softmax(dims, seed=3):
    """SoftmaxLastAxis over ONE row (no leading variables) and over ONE element (no variables at all): ops/softmax_last_axis has no is_scalar branch"""
    def g():
        n = int(np.prod(dims))
        nodes = [{"idx": 0, "op": "Input", "inputs": [], "dims": list(dims)},
                 {"idx": 1, "op": "SoftmaxLastAxis", "inputs": [0], "dims": list(dims), "scale": 14}]
        if n == 1:
            nodes += [{"idx": 2, "op": "Broadcast", "inputs": [1], "dims": [1, 2]},
                      {"idx": 3, "op": "Constant", "inputs": [], "dims": [1, 2], "data": np.array([1, 2], dtype=np.int32)},
                      {"idx": 4, "op": "Add", "inputs": [2, 3], "dims": [1, 2]}]
        return nodes, [nodes[-1]["idx"]], [np.random.default_rng(seed).integers(-(1 << 15), 1 << 15, size=n).astype(np.int32)]
    g.__name__ = "softmax_%s_graph" % "x".join(str(d) for d in dims)
    return g


SOFTMAX_ROWS = [_softmax([1, 8]), _softmax([1, 1, 16], 4), _softmax([1, 2], 6), _softmax([1, 4], 7)]
SOFTMAXES = [_softmax([1, 1]), _softmax([1, 1], 5)]

GRAPHS = GATHERS + SOFTMAX_ROWS + [scd_graph, scd_negative_graph, div_graph, einsum_scalar_graph, relu_graph, relu_negative_graph, clamp_graph,
          tanh_graph, erf_graph, sigmoid_graph, rsqrt_graph, sin_graph, cos_graph]


@pytest.mark.parametrize("builder", GRAPHS, ids=[b.__name__ for b in GRAPHS])
def test_one_element_proof_matches_oracle_and_verifies(atlas, builder):
    from oracle import graph as OG, orc
    from jolt_atlas_amd import graph as GG
    nodes, outputs, inputs = builder()
    tau = orc.random_fr(1, 0x51250001)[0]
    srs_h = orc.srs_powers(tau, 1 << 8)
    srs = atlas.SRS.generate(tau, 1 << 8)
    vk = atlas.HyperKZG.vk_from_trapdoor(tau, srs.download(0, 1)[0])
    P = OG.Prover(nodes, outputs, srs_h)
    want = P.prove(inputs)
    G = GG.Graph(nodes, outputs)
    try:
        G.trace(inputs)
        for nd in nodes:
            assert np.array_equal(G.node_output(nd["idx"]), P.trace[nd["idx"]]), f"trace of node {nd['idx']} ({nd['op']})"
        got, state, tm = G.prove(srs, inputs)
        assert state == P.t.state(), "final transcript state"
        assert got == want, "ONNXProof bytes"
        assert tm["n_committed"] == len(P.committed)
        out = G.node_output(outputs[0])
        V = GG.Graph(nodes, outputs)
        try:
            ok, vstate = V.verify(vk, inputs, out, got)
            assert ok and vstate == state
            bad = out.copy(); bad[0] += 1
            assert not V.verify(vk, inputs, bad, got)[0]
        finally:
            V.free()
    finally:
        G.free(); srs.free()


def random_one_element_chain(seed):
    """Input -> Sum -> Sum -> one element, then a seeded sequence of one-element operators (the is_scalar arithmetic of round 3 together with
    ScalarConstDiv and Div), then back to a vector"""
    rng = np.random.default_rng(seed)
    nodes = [{"idx": 0, "op": "Input", "inputs": [], "dims": [2, 2]},
             {"idx": 1, "op": "Sum", "inputs": [0], "dims": [2, 1], "axes": [1]},
             {"idx": 2, "op": "Sum", "inputs": [1], "dims": [1, 1], "axes": [0]}]
    cur = 2

    def add(op, inputs, **kw):
        nd = {"idx": len(nodes), "op": op, "inputs": list(inputs), "dims": [1, 1]}
        nd.update(kw); nodes.append(nd)
        return nd["idx"]

    def const(lo, hi):
        return add("Constant", [], data=np.array([int(rng.integers(lo, hi))], dtype=np.int32))

    for _ in range(int(rng.integers(3, 7))):
        r = rng.random()
        if r < 0.25:
            cur = add("ScalarConstDiv", [cur], divisor=int(rng.integers(2, 60)))
        elif r < 0.50:
            c = const(1, 1 << 10)                                   # a positive divisor
            cur = add("Div", [cur, c])
        elif r < 0.70:
            c = const(-(1 << 12), 1 << 12)
            cur = add(str(rng.choice(["Add", "Sub"])), [cur, c] if rng.random() < 0.5 else [c, cur])
        elif r < 0.80:
            c = const(-(1 << 8), 1 << 8)
            cur = add("Mul", [cur, c], scale=int(rng.integers(2, 6)))
        elif r < 0.88:
            cur = add("Square", [cur], scale=int(rng.integers(6, 12)))
        else:                                                       # (drawn from a second generator: the first eight seeds keep their chains up to here)
            op = str(np.random.default_rng(seed * 977 + len(nodes)).choice(["ReLU", "Clamp", "Tanh", "Sigmoid", "Erf", "Sin", "Cos"]))
            cur = add(op, [cur], **({"bound_log": 9} if op == "Clamp" else {} if op == "ReLU" else {"scale": 14}))
    last = cur
    nodes += [{"idx": len(nodes), "op": "Broadcast", "inputs": [last], "dims": [1, 2]}]
    nodes += [{"idx": len(nodes), "op": "Constant", "inputs": [], "dims": [1, 2], "data": rng.integers(-100, 100, size=2).astype(np.int32)}]
    nodes += [{"idx": len(nodes), "op": "Add", "inputs": [len(nodes) - 2, len(nodes) - 1], "dims": [1, 2]}]
    return nodes, [len(nodes) - 1], [rng.integers(-(1 << 10), 1 << 10, size=4).astype(np.int32)]


@pytest.mark.parametrize("seed", range(12))
def test_random_one_element_chain(atlas, seed):
    """the fuzz draw over the one-element operators: device proof == oracle proof, accepted by the verifier, one flipped byte rejected"""
    from oracle import graph as OG, orc
    from jolt_atlas_amd import graph as GG
    nodes, outputs, inputs = random_one_element_chain(seed)
    tau = orc.random_fr(1, 0x51250001)[0]
    srs_h = orc.srs_powers(tau, 1 << 8)
    srs = atlas.SRS.generate(tau, 1 << 8)
    vk = atlas.HyperKZG.vk_from_trapdoor(tau, srs.download(0, 1)[0])
    P = OG.Prover(nodes, outputs, srs_h)
    want = P.prove(inputs)
    G = GG.Graph(nodes, outputs)
    try:
        got, state, _ = G.prove(srs, inputs)
        for nd in nodes:
            assert np.array_equal(G.node_output(nd["idx"]), P.trace[nd["idx"]]), f"trace of node {nd['idx']} ({nd['op']})"
        assert state == P.t.state() and got == want, [n["op"] for n in nodes]
        out = G.node_output(outputs[0])
        V = GG.Graph(nodes, outputs)
        try:
            ok, vstate = V.verify(vk, inputs, out, got)
            assert ok and vstate == state
            flipped = bytearray(got); flipped[len(flipped) // 2] ^= 1
            assert not V.verify(vk, inputs, out, bytes(flipped))[0]
        finally:
            V.free()
    finally:
        G.free(); srs.free()


@pytest.mark.parametrize("builder", SOFTMAXES, ids=[b.__name__ for b in SOFTMAXES])
def test_softmax_over_one_element_is_refused_by_name(atlas, builder):
    """the last refusal: the device's softmax instances need rows of two elements (ONE row is composed — SOFTMAX_ROWS; the oracle composes ONE
    element too: its proof is what a device path would have to reproduce); the refusal names the operator when the graph is built, nothing is proved"""
    from oracle import graph as OG, orc
    from jolt_atlas_amd import graph as GG
    nodes, outputs, inputs = builder()
    srs_h = orc.srs_powers(orc.random_fr(1, 0x51250001)[0], 1 << 8)
    assert len(OG.Prover(nodes, outputs, srs_h).prove(inputs)) > 0
    with pytest.raises(atlas.AtlasError, match="SoftmaxLastAxis"):
        G = GG.Graph(nodes, outputs)
        try:
            G.trace(inputs)
        finally:
            G.free()


def random_one_cycle_graph(seed):
    """the seeded draw over the one-cycle gathers and one-row softmaxes (review item 8): a gather of ONE index from a dictionary of 2..256 words of
    1..8 elements (GatherSmall or GatherLarge), or SoftmaxLastAxis over ONE row of 2..16 elements, followed by an Add of a constant"""
    rng = np.random.default_rng(1000 + seed)
    if rng.random() < 0.6:
        op = str(rng.choice(["GatherSmall", "GatherLarge"]))
        V = int(2 ** rng.integers(1, 9)); word = int(2 ** rng.integers(0, 4))
        if op == "GatherLarge" and V < 16:
            V = 16
        idx = int(rng.integers(0, V))
        nodes = [{"idx": 0, "op": "Input", "inputs": [], "dims": [1]},
                 {"idx": 1, "op": "Constant", "inputs": [], "dims": [V, word], "data": rng.integers(-(1 << 14), 1 << 14, size=V * word).astype(np.int32)},
                 {"idx": 2, "op": op, "inputs": [1, 0], "dims": [1, word], "axis": 0, "dict_len": V}]
        inputs = [np.array([idx], dtype=np.int32)]
        width = word
    else:
        N = int(2 ** rng.integers(1, 5))
        nodes = [{"idx": 0, "op": "Input", "inputs": [], "dims": [1, N]},
                 {"idx": 1, "op": "SoftmaxLastAxis", "inputs": [0], "dims": [1, N], "scale": 14}]
        inputs = [rng.integers(-(1 << 15), 1 << 15, size=N).astype(np.int32)]
        width = N
    last = nodes[-1]["idx"]
    if width == 1:
        nodes += [{"idx": last + 1, "op": "Broadcast", "inputs": [last], "dims": [1, 2]}]
        last, width = last + 1, 2
    nodes += [{"idx": last + 1, "op": "Constant", "inputs": [], "dims": [1, width], "data": rng.integers(-9, 9, size=width).astype(np.int32)},
              {"idx": last + 2, "op": "Add", "inputs": [last, last + 1], "dims": [1, width]}]
    return nodes, [last + 2], inputs


@pytest.mark.parametrize("seed", range(8))
def test_random_one_cycle_graph(atlas, seed):
    from oracle import graph as OG, orc
    from jolt_atlas_amd import graph as GG
    nodes, outputs, inputs = random_one_cycle_graph(seed)
    tau = orc.random_fr(1, 0x51250001)[0]
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import build_graphs as BG
    nv = max(BG.max_vars(nodes), 8)
    srs_h = orc.srs_powers(tau, 1 << nv)
    srs = atlas.SRS.generate(tau, 1 << nv)
    vk = atlas.HyperKZG.vk_from_trapdoor(tau, srs.download(0, 1)[0])
    P = OG.Prover(nodes, outputs, srs_h)
    want = P.prove(inputs)
    G = GG.Graph(nodes, outputs)
    try:
        got, state, _ = G.prove(srs, inputs)
        for nd in nodes:
            assert np.array_equal(G.node_output(nd["idx"]), P.trace[nd["idx"]]), f"trace of node {nd['idx']} ({nd['op']})"
        assert state == P.t.state() and got == want, [n["op"] for n in nodes]
        out = G.node_output(outputs[0])
        V = GG.Graph(nodes, outputs)
        try:
            ok, vstate = V.verify(vk, inputs, out, got)
            assert ok and vstate == state
            flipped = bytearray(got); flipped[len(flipped) // 2] ^= 1
            assert not V.verify(vk, inputs, out, bytes(flipped))[0]
        finally:
            V.free()
    finally:
        G.free(); srs.free()
