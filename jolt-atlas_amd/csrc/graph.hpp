// The model graph, the execution trace and the prover state of ONNXProof::prove as the library holds them
// (jolt-atlas-core/src/onnx_proof/{mod,prover,types}.rs; atlas-onnx-tracer/src/{node/mod.rs,model/trace.rs};
// joltworks/src/poly/opening_proof.rs:1167-1320 OpeningId / SumcheckId; common/src/lib.rs CommittedPoly / VirtualPoly).
// Internal to libatlas_hip.so: graph_exec.hip (quantised execution, f3), graph_prove.hip (node loop, x1),
// graph_wire.hip (the ONNXProof container, f2), graph_verify.hip (ONNXProof::verify, f4).
#pragma once
#include <cstdint>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/atlas_hip.h"
#include "host_field.hpp"

namespace gr {
namespace H = atlas_host;

// ---- identifiers, ordered like the derived Ord of the Rust types -------------------------------------------------
// VirtualPoly (common/src/lib.rs), variant index = declaration order = the u8 tag of canonical_serde_enum!
enum VP : uint8_t {
    VP_NodeOutput = 0, VP_NodeOutputRa, VP_CosRa, VP_SinRa, VP_TrigDownscaleRa, VP_SoftmaxSumOutput, VP_SoftmaxMaxOutput, VP_SoftmaxMaxIndex,
    VP_HammingWeight, VP_DivRangeCheckRa, VP_SqrtRangeCheckRa, VP_TeleportRangeCheckRa, VP_MeanOfSquaresRangeCheckRa, VP_DivRemainder,
    VP_SqrtRemainder, VP_TeleportQuotient, VP_TeleportRemainder, VP_TrigDownscaled, VP_SoftmaxExpSum, VP_SoftmaxExpQ, VP_SoftmaxRemainderRa,
    VP_SoftmaxExpHi, VP_SoftmaxExpLo, VP_SoftmaxExpRemainder, VP_SoftmaxExpRemainderRa, VP_SoftmaxZHi, VP_SoftmaxZLo, VP_SoftmaxZHiRa,
    VP_SoftmaxZLoRa, VP_SoftmaxClampWitness, VP_SoftmaxClampRa, VP_SoftmaxRecipMultRemainder, VP_ClampAcc, VP_ClampRa, VP_RescaleRemainder,
    VP_RescaleRemainderRa, VP_SymmetricClampRa, VP_ActivationClampedOutput, VP_ActivationClampRa, VP_ActivationSmallRa, VP_COUNT
};
// number of usize fields of each VirtualPoly variant
inline int vp_arity(uint8_t v) { return v == VP_HammingWeight ? 0 : (v >= VP_SoftmaxSumOutput && v <= VP_SoftmaxMaxIndex) ? 2 : 1; }
// CommittedPoly
enum CP : uint8_t {
    CP_NodeOutputRaD = 0, CP_CosRaD, CP_SinRaD, CP_TrigDownscaleRaD, CP_DivRangeCheckRaD, CP_SqrtDivRangeCheckRaD, CP_MeanOfSquaresRangeCheckRaD,
    CP_SqrtRangeCheckRaD, CP_TeleportRangeCheckRaD, CP_DivNodeQuotient, CP_ScalarConstDivNodeRemainder, CP_RsqrtQuotient, CP_TeleportNodeQuotient,
    CP_GatherRa, CP_GatherRaD, CP_SoftmaxRemainderRaD, CP_SoftmaxExpRemainderRaD, CP_SoftmaxZHiRaD, CP_SoftmaxZLoRaD, CP_ClampRaD,
    CP_RescaleRemainderRaD, CP_SymmetricClampRaD, CP_ActivationClampRaD, CP_ActivationSmallRaD, CP_SoftmaxClampRaD, CP_COUNT
};
inline int cp_arity(uint8_t v) { return (v >= CP_DivNodeQuotient && v <= CP_GatherRa) ? 1 : 2; }
// SumcheckId
enum SC : uint8_t { SC_NodeExecution = 0, SC_Raf, SC_RaVirtualization, SC_RamHammingBooleanity, SC_RamHammingWeight, SC_Booleanity, SC_HammingWeight,
                    SC_RLC, SC_BlindFoldBatchOpening, SC_NTEvalShift };

struct PolyId {                 // PolynomialId: Virtual(VirtualPoly) < Committed(CommittedPoly)
    uint8_t committed = 0, var = 0;
    uint64_t a = 0, b = 0;
    bool operator<(const PolyId& o) const { return std::tie(committed, var, a, b) < std::tie(o.committed, o.var, o.a, o.b); }
    bool operator==(const PolyId& o) const { return committed == o.committed && var == o.var && a == o.a && b == o.b; }
};
inline PolyId virt(uint8_t v, uint64_t a = 0, uint64_t b = 0) { return PolyId{0, v, a, b}; }
inline PolyId comm(uint8_t v, uint64_t a = 0, uint64_t b = 0) { return PolyId{1, v, a, b}; }
struct OpeningId {
    PolyId poly;
    uint8_t sc = 0;
    uint64_t sc_idx = 0;       // NodeExecution / RLC payload
    bool operator<(const OpeningId& o) const { return std::tie(poly, sc, sc_idx) < std::tie(o.poly, o.sc, o.sc_idx); }
};
inline OpeningId oid(PolyId p, uint8_t sc, uint64_t idx = 0) { return OpeningId{p, sc, idx}; }
inline OpeningId node_exec(PolyId p, uint64_t node) { return OpeningId{p, SC_NodeExecution, node}; }

using Point = std::vector<H::Fr>;
struct Opening { Point point; H::Fr claim; };

// ProofType (jolt-atlas-core/src/onnx_proof/types.rs:34-75)
enum PT : uint8_t { PT_Execution = 0, PT_NeuralTeleport, PT_RaOneHotChecks, PT_RaHammingWeight, PT_RangeCheck, PT_SoftmaxStage1, PT_SoftmaxStage2,
                    PT_SoftmaxStage3, PT_SoftmaxStage4, PT_SumReduction, PT_EinsumMatmul, PT_RescaleRemainderRaChecks, PT_RescaleArith,
                    PT_TrigDownscaleRaChecks };
struct ProofId { uint64_t node; uint8_t type; bool operator<(const ProofId& o) const { return std::tie(node, type) < std::tie(o.node, o.type); } };

// ---- the graph ----------------------------------------------------------------------------------------------------
struct Node {
    size_t idx = 0;
    int op = 0;                              // ATLAS_OP_*
    std::vector<size_t> inputs, dims;        // output_dims
    int64_t p[6] = {0, 0, 0, 0, 0, 0};       // operator fields (scale, layout, axis, ...), see atlas_hip.h
    std::vector<size_t> shape;               // Broadcast / Reshape target, Sum axes, einsum dims
    std::vector<int32_t> constant;           // Constant(Tensor<i32>)
    size_t numel() const { size_t n = 1; for (size_t d : dims) n *= d; return n; }
};

// a committed witness polynomial (CommittedPoly -> MultilinearPolynomial): one-hot chunk of device-resident lookup
// indices, or a dense vector of small scalars
struct Committed {
    PolyId id;
    int kind = 1;                            // 1 one-hot, 0 dense
    size_t lkc = 4;                          // one-hot: log_K_chunk (4 = LOG_K_CHUNK of the RaD families; GatherRa of GatherSmall: the whole log_K)
    const uint64_t* d_lookups = nullptr;     // one-hot: T lookup indices in HBM (borrowed from the node witness)
    size_t log_T = 0, log_K = 0, chunk = 0;  // one-hot: chunk `chunk` of the log_K-bit index (OneHotParams::lookup_index_chunk)
    atlas_poly_t dense = nullptr;            // dense (owned by the node witness)
    atlas_g1_affine_t commitment;
    // the opening that reaches the reduction sumcheck (the last append for this polynomial)
    bool opened = false;
    Point point; H::Fr claim;
};

}  // namespace gr
