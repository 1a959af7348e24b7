// Wire formats: the ark-serialize 0.5 encodings the reference's proofs and SRS files use, so that
// bytes can be exchanged with a run of the reference.
//   Fr                    32 B little-endian canonical integer
//   G1Affine compressed   x as 32 B LE, flags in the two top bits of the last byte: bit 7 = y is the larger of
//                         (y, -y), bit 6 = point at infinity (SWFlags)               [SURVEY App. A.3]
//   G1Affine uncompressed x || y, the same flag bits on y's last byte
//   Vec<T>                u64 LE length, then the items
//   SumcheckInstanceProof { compressed_polys: Vec<CompressedUniPoly { coeffs_except_linear_term: Vec<Fr> }> }
//                         (joltworks/src/subprotocols/sumcheck.rs:624-640, poly/unipoly.rs:27-30)
//   HyperKZGProof { com: Vec<G1Affine>, w: Vec<G1Affine>, v: Vec<Vec<Fr>> }   (hyperkzg/mod.rs:172-177; 368 B at ell = 2,
//                         hyperkzg/tests.rs:107-109)
//   SRS { g1_powers: Vec<G1Affine>, g2_powers: Vec<G2Affine>, g_products: Vec<G1Affine> } compressed
//                         (hyperkzg/kzg.rs:18-23; HyperKZGSRS::load_from_file / save_to_file, hyperkzg/mod.rs:60-95)
// The flag-bit convention is the one SURVEY App. A.3 infers; it is pinned by tests/test_reference_fixtures.py when a
// fixture exported from the reference is present.  Decompressing an SRS is a square root per point: done on the
// device (2^24 points in ~60 ms) — one host core would need minutes.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <vector>

#include "../../include/atlas_hip.h"
#include "host_curve.hpp"
#include "host_field.hpp"
#include "runtime.hpp"
#include "srs.hpp"

using namespace atlas;
namespace H = atlas_host;
using atlas_rt::fail;
using atlas_rt::rt;

namespace {

// (p + 1) / 4 for Fq (p = 3 mod 4): sqrt(a) = a^((p+1)/4)
const uint64_t FQ_SQRT_EXP[4] = {0x4f082305b61f3f52ULL, 0x65e05aa45a1c72a3ULL, 0x6e14116da0605617ULL, 0x0c19139cb84c680aULL};
// (p - 1) / 2: y is "negative" (the larger of y, -y) iff canonical y > (p - 1) / 2
const uint64_t FQ_HALF[4] = {0x9e10460b6c3e7ea3ULL, 0xcbc0b548b438e546ULL, 0xdc2822db40c0ac2eULL, 0x183227397098d014ULL};

bool gt4(const uint64_t a[4], const uint64_t b[4]) {
    for (int i = 3; i >= 0; i--) if (a[i] != b[i]) return a[i] > b[i];
    return false;
}
H::Fq q_canon(const H::Fq& a) { return H::q_mul(a, H::Fq{{1, 0, 0, 0}}); }
H::Fq q_pow(const H::Fq& a, const uint64_t e[4]) {
    H::Fq acc = H::q_one(), base = a;
    for (int i = 0; i < 256; i++) { if ((e[i >> 6] >> (i & 63)) & 1) acc = H::q_mul(acc, base); base = H::q_sqr(base); }
    return acc;
}
void put_u64(uint8_t* p, uint64_t v) { for (int i = 0; i < 8; i++) p[i] = (uint8_t)(v >> (8 * i)); }
uint64_t get_u64(const uint8_t* p) { uint64_t v = 0; for (int i = 0; i < 8; i++) v |= (uint64_t)p[i] << (8 * i); return v; }

void fr_to_le(const atlas_fr_t* a, uint8_t out[32]) {
    uint64_t c[4];
    H::to_canonical(*reinterpret_cast<const H::Fr*>(a), c);
    for (int i = 0; i < 4; i++) put_u64(out + 8 * i, c[i]);
}
bool fr_from_le(const uint8_t in[32], atlas_fr_t* out) {
    uint64_t c[4];
    for (int i = 0; i < 4; i++) c[i] = get_u64(in + 8 * i);
    if (H::geq_p(c)) return false;
    const H::Fr f = H::from_canonical(c);
    std::memcpy(out, &f, 32);
    return true;
}
void g1_compress(const atlas_g1_affine_t* p, uint8_t out[32]) {
    H::Fq x, y;
    std::memcpy(x.l, p->x.l, 32); std::memcpy(y.l, p->y.l, 32);
    if (p->infinity || (H::q_is_zero(x) && H::q_is_zero(y))) { std::memset(out, 0, 32); out[31] = 0x40; return; }
    const H::Fq cx = q_canon(x), cy = q_canon(y);
    for (int i = 0; i < 4; i++) put_u64(out + 8 * i, cx.l[i]);
    if (gt4(cy.l, FQ_HALF)) out[31] |= 0x80;
}
int g1_decompress(const uint8_t in[32], atlas_g1_affine_t* out) {
    uint8_t b[32];
    std::memcpy(b, in, 32);
    const uint8_t flags = b[31] & 0xc0;
    b[31] &= 0x3f;
    std::memset(out, 0, sizeof(*out));
    if (flags & 0x40) {
        if (flags & 0x80) return fail(ATLAS_EINVAL, "G1 deserialize: infinity and sign flags both set");
        for (int i = 0; i < 32; i++) if (b[i]) return fail(ATLAS_EINVAL, "G1 deserialize: non-zero x with the infinity flag");
        out->infinity = 1;
        return ATLAS_OK;
    }
    uint64_t c[4];
    for (int i = 0; i < 4; i++) c[i] = get_u64(b + 8 * i);
    if (H::q_geq_p(c)) return fail(ATLAS_EINVAL, "G1 deserialize: x >= p");
    const H::Fq x = H::q_mul(H::Fq{{c[0], c[1], c[2], c[3]}}, H::Fq{{H::FQ_R2[0], H::FQ_R2[1], H::FQ_R2[2], H::FQ_R2[3]}});
    const H::Fq rhs = H::q_add(H::q_mul(H::q_sqr(x), x), H::q_from_u64(3));
    H::Fq y = q_pow(rhs, FQ_SQRT_EXP);
    if (!H::q_eq(H::q_sqr(y), rhs)) return fail(ATLAS_EINVAL, "G1 deserialize: x is not on the curve");
    const bool neg = gt4(q_canon(y).l, FQ_HALF);
    if (neg != ((flags & 0x80) != 0)) y = H::q_neg(y);
    std::memcpy(out->x.l, x.l, 32); std::memcpy(out->y.l, y.l, 32);
    return ATLAS_OK;
}

// ---- device: decompress an SRS
struct FqExp { uint32_t w[8]; };
__global__ __launch_bounds__(256) void k_g1_decompress(const uint8_t* __restrict__ in, size_t n, FqExp e, G1Affine* __restrict__ out, uint32_t* bad) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const uint32_t* w = reinterpret_cast<const uint32_t*>(in + 32 * i);
        Fe c;
        for (int k = 0; k < 8; k++) c.v[k] = w[k];
        const uint32_t flags = c.v[7] >> 30;
        c.v[7] &= 0x3fffffffu;
        G1Affine P;
        P.x = fe_zero(); P.y = fe_zero();
        if (flags & 1u) {                               // infinity
            if ((flags & 2u) || !fe_is_zero(c)) *bad = 1u;
        } else {
            const Fq x = fe_to_mont<FqParams>(c);       // (x >= p is caught by the curve check for all practical purposes; checked exactly below)
            bool ge = true;
            for (int k = 7; k >= 0; k--) if (c.v[k] != FqParams::p(k)) { ge = c.v[k] > FqParams::p(k); break; }
            Fq three = fq_one(); three = fq_add(fq_add(three, three), fq_one());
            const Fq rhs = fq_add(fq_mul(fq_sqr(x), x), three);
            Fq acc = fq_one(), base = rhs;
            for (int b = 0; b < 253; b++) { if ((e.w[b >> 5] >> (b & 31)) & 1u) acc = fq_mul(acc, base); base = fq_sqr(base); }
            Fq y = acc;
            if (ge || !fe_eq(fq_sqr(y), rhs)) *bad = 1u;
            const Fe cy = fe_from_mont<FqParams>(y);
            // y > (p - 1) / 2  <=>  2 y > p - 1  <=>  2 y >= p (p odd)
            Fe two_y; uint32_t carry = 0;
            for (int k = 0; k < 8; k++) { const uint64_t s = ((uint64_t)cy.v[k] << 1) | carry; two_y.v[k] = (uint32_t)s; carry = (uint32_t)(s >> 32); }
            bool neg = carry != 0;
            if (!neg) { neg = true; for (int k = 7; k >= 0; k--) if (two_y.v[k] != FqParams::p(k)) { neg = two_y.v[k] > FqParams::p(k); break; } }
            if (neg != ((flags & 2u) != 0)) y = fq_neg(y);
            P.x = x; P.y = y;
        }
        fe_store(&out[i].x, P.x); fe_store(&out[i].y, P.y);
    }
}

__global__ __launch_bounds__(256) void k_g1_compress(const G1Affine* __restrict__ in, size_t n, uint8_t* __restrict__ out) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const G1Affine P = g1_aff_load(in + i);
        uint32_t* w = reinterpret_cast<uint32_t*>(out + 32 * i);
        if (g1_aff_is_inf(P)) { for (int k = 0; k < 7; k++) w[k] = 0; w[7] = 0x40000000u; continue; }
        const Fe cx = fe_from_mont<FqParams>(P.x), cy = fe_from_mont<FqParams>(P.y);
        uint32_t carry = 0; Fe two_y;
        for (int k = 0; k < 8; k++) { const uint64_t s = ((uint64_t)cy.v[k] << 1) | carry; two_y.v[k] = (uint32_t)s; carry = (uint32_t)(s >> 32); }
        bool neg = carry != 0;
        if (!neg) { neg = true; for (int k = 7; k >= 0; k--) if (two_y.v[k] != FqParams::p(k)) { neg = two_y.v[k] > FqParams::p(k); break; } }
        for (int k = 0; k < 8; k++) w[k] = cx.v[k];
        if (neg) w[7] |= 0x80000000u;
    }
}

}  // namespace

extern "C" {

int atlas_fr_to_bytes(const atlas_fr_t* a, uint8_t out[32]) {
    if (!a || !out) return fail(ATLAS_EINVAL, "fr_to_bytes");
    fr_to_le(a, out);
    return ATLAS_OK;
}
int atlas_fr_from_bytes(const uint8_t in[32], atlas_fr_t* out) {
    if (!in || !out) return fail(ATLAS_EINVAL, "fr_from_bytes");
    if (!fr_from_le(in, out)) return fail(ATLAS_EINVAL, "Fr deserialize: value >= r");
    return ATLAS_OK;
}
int atlas_g1_to_bytes_compressed(const atlas_g1_affine_t* p, uint8_t out[32]) {
    if (!p || !out) return fail(ATLAS_EINVAL, "g1_to_bytes_compressed");
    g1_compress(p, out);
    return ATLAS_OK;
}
int atlas_g1_from_bytes_compressed(const uint8_t in[32], atlas_g1_affine_t* out) {
    if (!in || !out) return fail(ATLAS_EINVAL, "g1_from_bytes_compressed");
    return g1_decompress(in, out);
}
int atlas_g1_to_bytes_uncompressed(const atlas_g1_affine_t* p, uint8_t out[64]) {
    if (!p || !out) return fail(ATLAS_EINVAL, "g1_to_bytes_uncompressed");
    H::Fq x, y;
    std::memcpy(x.l, p->x.l, 32); std::memcpy(y.l, p->y.l, 32);
    if (p->infinity || (H::q_is_zero(x) && H::q_is_zero(y))) { std::memset(out, 0, 64); out[63] = 0x40; return ATLAS_OK; }
    const H::Fq cx = q_canon(x), cy = q_canon(y);
    for (int i = 0; i < 4; i++) { put_u64(out + 8 * i, cx.l[i]); put_u64(out + 32 + 8 * i, cy.l[i]); }
    if (gt4(cy.l, FQ_HALF)) out[63] |= 0x80;
    return ATLAS_OK;
}

// SumcheckInstanceProof::serialize_compressed: rows[i * row_stride ..] = the n_coeffs[i] compressed coefficients of round i
int atlas_sumcheck_proof_serialize(const atlas_fr_t* rows, size_t row_stride, const uint32_t* n_coeffs, size_t n_rounds, uint8_t* out,
                                   size_t cap, size_t* len) {
    if ((n_rounds && (!rows || !n_coeffs)) || !len) return fail(ATLAS_EINVAL, "sumcheck_proof_serialize: null argument");
    size_t need = 8;
    for (size_t i = 0; i < n_rounds; i++) { if (n_coeffs[i] > row_stride) return fail(ATLAS_EINVAL, "sumcheck_proof_serialize: n_coeffs > row_stride"); need += 8 + 32 * (size_t)n_coeffs[i]; }
    *len = need;
    if (!out) return ATLAS_OK;                       // size query
    if (cap < need) return fail(ATLAS_EINVAL, "sumcheck_proof_serialize: buffer too small");
    uint8_t* p = out;
    put_u64(p, n_rounds); p += 8;
    for (size_t i = 0; i < n_rounds; i++) {
        put_u64(p, n_coeffs[i]); p += 8;
        for (uint32_t k = 0; k < n_coeffs[i]; k++, p += 32) fr_to_le(&rows[i * row_stride + k], p);
    }
    return ATLAS_OK;
}
int atlas_sumcheck_proof_deserialize(const uint8_t* in, size_t len, atlas_fr_t* rows, size_t row_stride, uint32_t* n_coeffs, size_t max_rounds,
                                     size_t* n_rounds, size_t* consumed) {
    if (!in || !rows || !n_coeffs || !n_rounds) return fail(ATLAS_EINVAL, "sumcheck_proof_deserialize: null argument");
    size_t o = 0;
    if (len < 8) return fail(ATLAS_EINVAL, "sumcheck_proof_deserialize: truncated");
    const uint64_t n = get_u64(in); o = 8;
    if (n > max_rounds) return fail(ATLAS_EINVAL, "sumcheck_proof_deserialize: more rounds than the caller allows");
    for (uint64_t i = 0; i < n; i++) {
        if (len < o + 8) return fail(ATLAS_EINVAL, "sumcheck_proof_deserialize: truncated");
        const uint64_t k = get_u64(in + o); o += 8;
        if (k > row_stride || len < o + 32 * k) return fail(ATLAS_EINVAL, "sumcheck_proof_deserialize: row too long or truncated");
        n_coeffs[i] = (uint32_t)k;
        for (uint64_t j = 0; j < k; j++, o += 32)
            if (!fr_from_le(in + o, &rows[i * row_stride + j])) return fail(ATLAS_EINVAL, "sumcheck_proof_deserialize: coefficient >= r");
    }
    *n_rounds = (size_t)n;
    if (consumed) *consumed = o;
    return ATLAS_OK;
}

// HyperKZGProof::serialize_compressed: com (n_com = ell - 1 points), w (3 points), v[i * ell + j] (3 rows of ell)
int atlas_hyperkzg_proof_serialize(const atlas_g1_affine_t* com, size_t n_com, const atlas_g1_affine_t* w, const atlas_fr_t* v, size_t ell,
                                   uint8_t* out, size_t cap, size_t* len) {
    if ((n_com && !com) || !w || !v || !len || ell == 0) return fail(ATLAS_EINVAL, "hyperkzg_proof_serialize: null argument");
    const size_t need = 8 + 32 * n_com + 8 + 32 * 3 + 8 + 3 * (8 + 32 * ell);
    *len = need;
    if (!out) return ATLAS_OK;
    if (cap < need) return fail(ATLAS_EINVAL, "hyperkzg_proof_serialize: buffer too small");
    uint8_t* p = out;
    put_u64(p, n_com); p += 8;
    for (size_t i = 0; i < n_com; i++, p += 32) g1_compress(&com[i], p);
    put_u64(p, 3); p += 8;
    for (size_t i = 0; i < 3; i++, p += 32) g1_compress(&w[i], p);
    put_u64(p, 3); p += 8;
    for (size_t i = 0; i < 3; i++) {
        put_u64(p, ell); p += 8;
        for (size_t j = 0; j < ell; j++, p += 32) fr_to_le(&v[i * ell + j], p);
    }
    return ATLAS_OK;
}
int atlas_hyperkzg_proof_deserialize(const uint8_t* in, size_t len, atlas_g1_affine_t* com, size_t cap_com, size_t* n_com, atlas_g1_affine_t* w,
                                     atlas_fr_t* v, size_t cap_ell, size_t* ell, size_t* consumed) {
    if (!in || !com || !n_com || !w || !v || !ell) return fail(ATLAS_EINVAL, "hyperkzg_proof_deserialize: null argument");
    size_t o = 0;
    auto need = [&](size_t k) { return len >= o + k; };
    if (!need(8)) return fail(ATLAS_EINVAL, "hyperkzg_proof_deserialize: truncated");
    const uint64_t nc = get_u64(in); o += 8;
    if (nc > cap_com || !need(32 * nc)) return fail(ATLAS_EINVAL, "hyperkzg_proof_deserialize: com too long or truncated");
    for (uint64_t i = 0; i < nc; i++, o += 32) { int rc = g1_decompress(in + o, &com[i]); if (rc) return rc; }
    if (!need(8) || get_u64(in + o) != 3) return fail(ATLAS_EINVAL, "hyperkzg_proof_deserialize: w must hold 3 points");
    o += 8;
    if (!need(96)) return fail(ATLAS_EINVAL, "hyperkzg_proof_deserialize: truncated");
    for (int i = 0; i < 3; i++, o += 32) { int rc = g1_decompress(in + o, &w[i]); if (rc) return rc; }
    if (!need(8) || get_u64(in + o) != 3) return fail(ATLAS_EINVAL, "hyperkzg_proof_deserialize: v must hold 3 rows");
    o += 8;
    uint64_t l0 = 0;
    for (int i = 0; i < 3; i++) {
        if (!need(8)) return fail(ATLAS_EINVAL, "hyperkzg_proof_deserialize: truncated");
        const uint64_t l = get_u64(in + o); o += 8;
        if (i == 0) l0 = l;
        if (l != l0 || l > cap_ell || !need(32 * l)) return fail(ATLAS_EINVAL, "hyperkzg_proof_deserialize: ragged or oversized v");
        for (uint64_t j = 0; j < l; j++, o += 32)
            if (!fr_from_le(in + o, &v[i * l + j])) return fail(ATLAS_EINVAL, "hyperkzg_proof_deserialize: evaluation >= r");
    }
    *n_com = (size_t)nc; *ell = (size_t)l0;
    if (consumed) *consumed = o;
    return ATLAS_OK;
}

// HyperKZGSRS::load_from_file (hyperkzg/mod.rs:60-67): reads g1_powers (the prover key) of a compressed SRS; max_points = 0 keeps all
int atlas_srs_load_file(const char* path, size_t max_points, atlas_srs_t* out) {
    NEED_INIT();
    if (!path || !out) return fail(ATLAS_EINVAL, "srs_load_file: null argument");
    FILE* f = std::fopen(path, "rb");
    if (!f) return fail(ATLAS_EINVAL, "srs_load_file: cannot open the file");
    uint8_t hdr[8];
    if (std::fread(hdr, 1, 8, f) != 8) { std::fclose(f); return fail(ATLAS_EINVAL, "srs_load_file: truncated"); }
    const uint64_t n_file = get_u64(hdr);
    // the header's length is untrusted: it must fit the file (and the multiplication below)
    long file_size = -1;
    if (std::fseek(f, 0, SEEK_END) == 0) file_size = std::ftell(f);
    if (file_size < 8 || std::fseek(f, 8, SEEK_SET) != 0 || n_file > ((uint64_t)file_size - 8) / 32) { std::fclose(f); return fail(ATLAS_EINVAL, "srs_load_file: the g1_powers length in the header exceeds the file"); }
    const size_t n = max_points && max_points < n_file ? max_points : (size_t)n_file;
    if (n == 0) { std::fclose(f); return fail(ATLAS_EINVAL, "srs_load_file: empty g1_powers"); }
    std::vector<uint8_t> buf;
    try { buf.resize(n * 32); } catch (const std::exception&) { std::fclose(f); return fail(ATLAS_ENOMEM, "srs_load_file: host buffer"); }
    const size_t got = std::fread(buf.data(), 1, buf.size(), f);
    std::fclose(f);
    if (got != buf.size()) return fail(ATLAS_EINVAL, "srs_load_file: truncated");
    std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
    atlas_srs* s = new atlas_srs();
    uint8_t* d_in = nullptr; uint32_t* d_bad = nullptr;
    hipError_t e = hipMalloc(&s->d, n * sizeof(G1Affine));
    if (e == hipSuccess) e = hipMalloc(&d_in, buf.size());
    if (e == hipSuccess) e = hipMalloc(&d_bad, 4);
    if (e == hipSuccess) e = hipMemsetAsync(d_bad, 0, 4, rt().stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_in, buf.data(), buf.size(), hipMemcpyHostToDevice, rt().stream);
    uint32_t bad = 0;
    if (e == hipSuccess) {
        FqExp ex;
        for (int k = 0; k < 4; k++) { ex.w[2 * k] = (uint32_t)FQ_SQRT_EXP[k]; ex.w[2 * k + 1] = (uint32_t)(FQ_SQRT_EXP[k] >> 32); }
        size_t gb = (n + 255) / 256; if (gb > 8192) gb = 8192;
        k_g1_decompress<<<(unsigned)gb, 256, 0, rt().stream>>>(d_in, n, ex, s->d, d_bad);
        e = hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, rt().stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(rt().stream);
    if (d_in) (void)hipFree(d_in);
    if (d_bad) (void)hipFree(d_bad);
    if (e != hipSuccess || bad) {
        if (s->d) (void)hipFree(s->d);
        delete s;
        return e != hipSuccess ? fail(ATLAS_ENODEV, "srs_load_file", e) : fail(ATLAS_EINVAL, "srs_load_file: a point is not a valid compressed G1Affine");
    }
    s->len = n;
    *out = s;
    return ATLAS_OK;
}

// writes SRS { g1_powers, g2_powers: [], g_products: [] } compressed: the prover half of HyperKZGSRS::save_to_file
int atlas_srs_save_file(atlas_srs_t srs, const char* path) {
    NEED_INIT();
    if (!srs || !path) return fail(ATLAS_EINVAL, "srs_save_file: null argument");
    std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
    std::vector<uint8_t> buf(srs->len * 32);
    uint8_t* d_out = nullptr;
    hipError_t e = hipMalloc(&d_out, buf.size());
    if (e != hipSuccess) return fail(ATLAS_ENOMEM, "hipMalloc(srs_save_file)", e);
    size_t gb = (srs->len + 255) / 256; if (gb > 8192) gb = 8192;
    k_g1_compress<<<(unsigned)gb, 256, 0, rt().stream>>>(srs->d, srs->len, d_out);
    e = hipMemcpyAsync(buf.data(), d_out, buf.size(), hipMemcpyDeviceToHost, rt().stream);
    if (e == hipSuccess) e = hipStreamSynchronize(rt().stream);
    (void)hipFree(d_out);
    if (e != hipSuccess) return fail(ATLAS_ENODEV, "srs_save_file", e);
    FILE* f = std::fopen(path, "wb");
    if (!f) return fail(ATLAS_EINVAL, "srs_save_file: cannot create the file");
    uint8_t hdr[8];
    put_u64(hdr, srs->len);
    bool ok = std::fwrite(hdr, 1, 8, f) == 8 && std::fwrite(buf.data(), 1, buf.size(), f) == buf.size();
    put_u64(hdr, 0);
    ok = ok && std::fwrite(hdr, 1, 8, f) == 8 && std::fwrite(hdr, 1, 8, f) == 8;
    ok = std::fclose(f) == 0 && ok;
    return ok ? ATLAS_OK : fail(ATLAS_EINVAL, "srs_save_file: write failed");
}

}  // extern "C"
