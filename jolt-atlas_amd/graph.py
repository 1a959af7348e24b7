"""ONNXProof::prove over a model graph (atlas_graph_* / atlas_prove_graph): ctypes marshalling only.

A graph description is a plain list of node dicts in the tracer's vocabulary
(atlas-onnx-tracer/src/node/mod.rs:12-24, ops/mod.rs:117-155):
    {"idx": 3, "op": "Einsum", "inputs": [1, 2], "dims": [16, 64], "layout": "mk,kn->mn", "scale": 7, "shape": [16, 32, 64]}
`tools/build_graphs.py` builds the committed descriptions; `oracle/graph.py` proves the same description on the CPU."""
import ctypes as C

import numpy as np

from . import TranscriptState, _check, lib

OPS = ["Input", "Constant", "Identity", "Add", "Sub", "Mul", "Square", "Cube", "And", "Iff", "ReLU", "Einsum", "Reshape", "MoveAxis", "Broadcast",
       "Slice", "Concat", "Sum", "ScalarConstDiv", "Div", "MeanOfSquares", "Rsqrt", "SoftmaxLastAxis", "Tanh", "GatherLarge", "GatherSmall", "Erf", "Sigmoid", "Neg", "IsNan", "Clamp", "Sin", "Cos"]
OP = {n: i for i, n in enumerate(OPS)}
LAYOUTS = {"mk,kn->mn": 0, "bmk,bkn->mbn": 1, "bmk,kbn->mbn": 2, "mbk,bnk->bmn": 3, "mbk,nbk->bmn": 4, "k,nk->n": 5}


class GraphTiming(C.Structure):
    _fields_ = [("trace_ms", C.c_double), ("commit_ms", C.c_double), ("iop_ms", C.c_double), ("reduction_ms", C.c_double),
                ("hyperkzg_ms", C.c_double), ("total_ms", C.c_double), ("n_nodes", C.c_size_t), ("n_committed", C.c_size_t),
                ("n_sumchecks", C.c_size_t)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


def node_params(nd):
    """(params, shape) of a node dict as atlas_graph_add_node takes them."""
    op = nd["op"]
    if op in ("Mul", "Square", "Cube"):
        return [nd["scale"]], []
    if op == "Einsum":
        return [LAYOUTS[nd["layout"]], nd["scale"]], list(nd["shape"])
    if op == "MoveAxis":
        return [nd["source"], nd["destination"]], []
    if op == "Slice":
        return [nd["axis"], nd["start"], nd["end"]], []
    if op == "Concat":
        return [nd["axis"]], []
    if op == "Clamp":
        return [nd["bound_log"]], []
    if op == "ScalarConstDiv":
        return [nd["divisor"]], []
    if op == "Sum":
        return [], list(nd["axes"])
    if op in ("Rsqrt", "SoftmaxLastAxis", "Tanh", "Erf", "Sigmoid", "Sin", "Cos"):
        return [nd["scale"]], []
    if op == "MeanOfSquares":
        return [nd["scale"], nd["count"]], list(nd["axes"])
    if op in ("GatherLarge", "GatherSmall"):
        return [nd["axis"], nd["dict_len"]], []
    return [], []


class Graph:
    def __init__(self, nodes, outputs):
        self.h = C.c_void_p()
        _check(lib.atlas_graph_new(C.byref(self.h)))
        self.nodes = nodes
        for nd in nodes:
            ins = (C.c_size_t * max(len(nd["inputs"]), 1))(*nd["inputs"])
            dims = (C.c_size_t * len(nd["dims"]))(*nd["dims"])
            params, shape = node_params(nd)
            pa = (C.c_int64 * max(len(params), 1))(*params)
            sh = (C.c_size_t * max(len(shape), 1))(*shape)
            const = None
            if nd["op"] == "Constant":
                cdata = np.ascontiguousarray(nd["data"], dtype=np.int32).reshape(-1)
                const = cdata.ctypes.data_as(C.POINTER(C.c_int32))
            _check(lib.atlas_graph_add_node(self.h, C.c_size_t(nd["idx"]), C.c_int(OP[nd["op"]]), ins, C.c_size_t(len(nd["inputs"])), dims,
                                            C.c_size_t(len(nd["dims"])), pa, C.c_size_t(len(params)), sh, C.c_size_t(len(shape)), const))
        o = (C.c_size_t * len(outputs))(*outputs)
        _check(lib.atlas_graph_set_outputs(self.h, o, C.c_size_t(len(outputs))))

    def _inputs(self, inputs):
        arrs = [np.ascontiguousarray(x, dtype=np.int32).reshape(-1) for x in inputs]
        # the C entry points read padded_len(dims) words from each pointer: a short array would be an out-of-bounds host read
        in_nodes = [nd for nd in self.nodes if nd["op"] == "Input"]
        if len(arrs) != len(in_nodes):
            raise ValueError(f"{len(in_nodes)} input tensors expected (one per Input node), got {len(arrs)}")
        for a, nd in zip(arrs, in_nodes):
            if a.size != int(np.prod(nd["dims"])):
                raise ValueError(f"input of node {nd['idx']}: {int(np.prod(nd['dims']))} elements expected for dims {nd['dims']}, got {a.size}")
        ptrs = (C.POINTER(C.c_int32) * max(len(arrs), 1))(*[a.ctypes.data_as(C.POINTER(C.c_int32)) for a in arrs])
        return arrs, ptrs

    def trace(self, inputs):
        arrs, ptrs = self._inputs(inputs)
        _check(lib.atlas_graph_trace(self.h, ptrs, C.c_size_t(len(arrs))))

    def node_output(self, idx):
        n = C.c_size_t()
        _check(lib.atlas_graph_node_output(self.h, C.c_size_t(idx), None, C.c_size_t(0), C.byref(n)))
        out = np.zeros(n.value, dtype=np.int32)
        _check(lib.atlas_graph_node_output(self.h, C.c_size_t(idx), out.ctypes.data_as(C.POINTER(C.c_int32)), n, C.byref(n)))
        return out

    def prove(self, srs, inputs, group=None):
        """ONNXProof::prove.  Returns (proof bytes, final transcript state bytes, timing dict).  group: a sharded.ShardGroup — every rank of
        it makes the same call (atlas_prove_graph_sharded: commitments by polynomial range, the opening's MSMs by point range) and gets
        the same bytes as one GPU does."""
        arrs, ptrs = self._inputs(inputs)
        n = C.c_size_t(); tm = GraphTiming(); ts = TranscriptState()
        cap = getattr(self, "_proof_cap", 1 << 22)      # a second proof of the same graph starts from the first one's size
        while True:
            buf = (C.c_uint8 * cap)()
            if group is None:
                rc = lib.atlas_prove_graph(self.h, srs.h, ptrs, C.c_size_t(len(arrs)), buf, C.c_size_t(cap), C.byref(n), C.byref(ts), C.byref(tm))
            else:
                rc = lib.atlas_prove_graph_sharded(self.h, srs.h, group.h, ptrs, C.c_size_t(len(arrs)), buf, C.c_size_t(cap), C.byref(n), C.byref(ts), C.byref(tm))
            if rc != 0 and n.value > cap:
                cap = n.value
                continue
            _check(rc)
            break
        self._proof_cap = max(cap, 1 << 22)
        return bytes(buf[:n.value]), bytes(ts.state), tm.as_dict()

    def verify(self, vk, inputs, output, proof):
        """ONNXProof::verify against the model, the inputs and the claimed output tensor.  True = accept, False = ProofVerifyError;
        raises for a graph with an operator the verifier does not compose.  Returns (accepted, final transcript state)."""
        arrs, ptrs = self._inputs(inputs)
        out = np.ascontiguousarray(output, dtype=np.int32).reshape(-1)
        buf = (C.c_uint8 * len(proof)).from_buffer_copy(proof)
        ts = TranscriptState()
        rc = lib.atlas_verify_graph(self.h, vk.ctypes.data_as(C.c_void_p), ptrs, C.c_size_t(len(arrs)), out.ctypes.data_as(C.POINTER(C.c_int32)),
                                    C.c_size_t(len(out)), buf, C.c_size_t(len(proof)), C.byref(ts))
        if rc == -5:                                   # ATLAS_EVERIFY
            return False, None
        _check(rc)
        return True, bytes(ts.state)

    def free(self):
        if self.h:
            lib.atlas_graph_free(self.h)
            self.h = C.c_void_p()
