// atlas_graph: the model (ComputationGraph), its execution trace in HBM (Trace: node index -> Tensor<i32>) and the
// per-node witness artifacts the prover derives from it (accumulators, quotients / remainders, lookup indices) —
// generate_node_witnesses (jolt-atlas-core/src/onnx_proof/witness.rs:136-200) keeps them only for the commitment; here
// they stay resident for the IOP so that nothing is executed twice.
#pragma once
#include "graph.hpp"
#include "node_flow.hip.h"

// SoftmaxLastAxisTrace (atlas-onnx-tracer/src/ops/softmax.rs:24-66) in HBM: [F][N] i32 tensors, the per-row vectors and the five
// lookup-index vectors of the node's one-hot families
struct SoftmaxWitness {
    size_t F = 0, N = 0, log2_base = 0, lk_hi = 0, lk_lo = 0;
    DevBuf R, exp_q, exp_hi, exp_lo, r_exp, z, z_hi, z_lo, e;           // i32 [F][N]; e = the argmax indicator (max.rs:70-74)
    DevBuf max_k, argmax_k, exp_sum, inv_sum;                             // i32 [F]
    DevBuf idx_R, idx_rexp, idx_z, idx_zhi, idx_zlo;                      // u64 [F][N]
};

struct NodeWitness {
    std::unique_ptr<RescaleWitness> rescale;     // Einsum / Mul / Square / Cube (fused rescale)
    std::unique_ptr<SoftmaxWitness> softmax;     // SoftmaxLastAxis
    DevBuf acc, acc_fr, cidx;                    // Add / Sub / Sum: i64 accumulation, its Fr image, the clamp lookup indices
    DevBuf lookups;                              // ReLU and the other XLEN-bit unary lookups; Div / MeanOfSquares: the range check's interleaved pairs
    DevBuf rem;                                  // ScalarConstDiv / Div: the remainder tensor (i32); Rsqrt: div_remainder
    DevBuf quot_fr, rem2, lookups2, bound;       // Rsqrt: quotient as Fr, sqrt_remainder, the second range check's pairs and its bound 2 out + 1
    DevBuf clamped;                              // Tanh: clamp(input, ACTIVATION_BOUND) (i32); its 18-bit lookups in `lookups2`, the raw-input lookups in `lookups`
    std::vector<std::unique_ptr<DevBuf>> one_cycle_rows;    // a lookup operator over ONE element: its K x 1 one-hot chunk polynomials as 16-coefficient dense rows
    std::vector<atlas_poly_t> dense_views;       // borrowed polynomial views of the dense committed polynomials
    ~NodeWitness() { for (auto p : dense_views) if (p) atlas_poly_free(p); }
    NodeWitness() = default;
    NodeWitness(const NodeWitness&) = delete;
    NodeWitness& operator=(const NodeWitness&) = delete;
    std::vector<gr::Committed> committed;        // this node's committed polynomials, in get_committed_polynomials order
};

struct atlas_graph {
    std::map<size_t, gr::Node> nodes;
    std::vector<size_t> outputs;
    // trace
    bool traced = false;
    std::map<size_t, DevBuf> out;                // node outputs, padded_next_power_of_two, i32
    std::map<size_t, NodeWitness> wit;
    DevBuf trace_err;                            // one word the trace's Gather kernels raise on an index outside the dictionary (graph_exec.hip)
    std::vector<size_t> input_nodes() const { std::vector<size_t> v; for (auto& kv : nodes) if (kv.second.op == ATLAS_OP_INPUT) v.push_back(kv.first); return v; }
    const int32_t* tensor(size_t idx) const { auto it = out.find(idx); return it == out.end() ? nullptr : it->second.as<int32_t>(); }
    void clear_trace() { out.clear(); wit.clear(); traced = false; }
    // a new trace of the same model: the Constant nodes' tensors (the weights) stay in HBM, everything derived from the inputs goes
    void clear_trace_keep_constants() {
        for (auto it = out.begin(); it != out.end();) { auto nd = nodes.find(it->first); if (nd != nodes.end() && nd->second.op == ATLAS_OP_CONSTANT) ++it; else it = out.erase(it); }
        wit.clear(); traced = false;
    }
};

// what exec_node would refuse, from the description alone (graph_exec.hip): called per node by the trace and, over the whole graph, by
// atlas_prove_graph and atlas_verify_graph before they read a node's dims / parameters
int atlas_rt_validate_node(const atlas_graph& G, const gr::Node& nd);
int atlas_rt_validate_graph(const atlas_graph& G);

// the small activation tables of Tanh / Erf / Sigmoid (ops/tanh.rs:22-32, erf.rs:22-32, sigmoid.rs:22-32 -> neural_teleport/utils.rs:67-85):
// Table[i] = round(2^14 f(signed18(i) / 2^14)), built once on the host (the reference's f64 arithmetic) and kept in HBM; graph_exec.hip.
// op = ATLAS_OP_TANH / ATLAS_OP_ERF / ATLAS_OP_SIGMOID
int atlas_rt_activation_table(int op, const int32_t** d_table, const std::vector<int32_t>** h_table);
// Sin / Cos by neural teleportation (common/src/consts/trig.rs, neural_teleport/{sin,cos}.rs): the period modulus round(k 2 pi 2^14) for the
// smallest k within the tolerance (k = 24: 2470649, 22 bits), remainders shifted down by 6 bits into a 2^16-entry table
// Table[i] = round(2^8 f(i / 2^8)) * 2^6
namespace gr { constexpr int64_t TRIG_PERIOD_MODULUS = 2470649; constexpr size_t TRIG_DOWNSCALE_BITS = 6, TRIG_TABLE_VARS = 16; }
int atlas_rt_trig_table(int op, const int32_t** d_table, const std::vector<int32_t>** h_table);
inline bool atlas_rt_is_activation(int op) { return op == ATLAS_OP_TANH || op == ATLAS_OP_ERF || op == ATLAS_OP_SIGMOID; }
// the decomposed exp sub-tables of SoftmaxLastAxis at MODEL_SCALE (generate_exp_lut_decomposed, atlas-onnx-tracer/src/ops/softmax.rs:239-269;
// lut_hi zero-padded to a power of two, :94-96), built once on the host in the reference's f64 arithmetic; graph_exec.hip
struct ExpLut { std::vector<int32_t> hi, lo; size_t log2_base = 0; const int32_t *d_hi = nullptr, *d_lo = nullptr; };
int atlas_rt_exp_lut(const ExpLut** out);

namespace gr {
constexpr size_t MODEL_SCALE = 14, ACTIVATION_BOUND = MODEL_SCALE + 3, ACTIVATION_TABLE_VARS = ACTIVATION_BOUND + 1;      // common/src/consts
constexpr size_t CLAMP_BOUND = 9;                                // the ONNX Clamp op's table (joltworks/src/lookup_tables/clamp.rs:197-211)
inline size_t next_pow2(size_t x) { size_t p = 1; while (p < x) p <<= 1; return p; }
inline unsigned log2u(size_t x) { unsigned n = 0; while (x > 1) { x >>= 1; n++; } return n; }
// pow2_padded_num_output_elements (node/mod.rs:52-57): every dimension padded on its own
inline size_t padded_len(const std::vector<size_t>& dims) { size_t n = 1; for (size_t d : dims) n *= next_pow2(d); return n; }
inline bool all_pow2(const std::vector<size_t>& dims) { for (size_t d : dims) if (d == 0 || (d & (d - 1))) return false; return true; }
}  // namespace gr
