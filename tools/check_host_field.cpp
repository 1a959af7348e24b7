// Cross-check of the host field multiplication (csrc/host_field.hpp): the unrolled CIOS form against the product-then-reduce
// form on 2.2 million operand pairs (random, near p, squares, and unreduced left operands < 2^256 as sum_to_fr feeds them),
// then its latency.  Built and run by tests/test_host_field.py:  g++ -O2 -std=c++17 -I. tools/check_host_field.cpp
#include <chrono>
#include <cstdio>
#include <random>
#include "jolt-atlas_amd/csrc/host_field.hpp"
namespace H = atlas_host;
int main() {
    std::mt19937_64 rng(1);
    size_t bad = 0;
    for (int i = 0; i < 2000000; i++) {
        H::Fr a{{rng(), rng(), rng(), rng() >> 2}}, b{{rng(), rng(), rng(), rng() >> 2}};
        if (i % 3 == 0) { a.l[3] = H::FR_P[3]; a.l[2] = H::FR_P[2]; a.l[1] = H::FR_P[1]; a.l[0] = H::FR_P[0] - 1 - (i & 7); }   // near p
        if (i % 5 == 0) { b = a; }
        if (H::geq_p(a.l)) H::sub_p(a.l);
        if (H::geq_p(b.l)) H::sub_p(b.l);
        H::Fr x = H::mul(a, b), y = H::mul_sos(a, b);
        if (!(x == y)) bad++;
    }
    // unreduced inputs < 2^256 as sum_to_fr feeds them (one operand < p)
    for (int i = 0; i < 200000; i++) {
        H::Fr a{{rng(), rng(), rng(), rng()}}, b{{rng(), rng(), rng(), rng() >> 3}};
        if (H::geq_p(b.l)) H::sub_p(b.l);
        if (!(H::mul(a, b) == H::mul_sos(a, b))) bad++;
    }
    // a sparse left operand (two low limbs zero: a 125-bit challenge in the upper half): the two-step form
    for (int i = 0; i < 200000; i++) {
        H::Fr r{{0, 0, rng(), rng() & 0x1fffffffffffffffULL}}, x{{rng(), rng(), rng(), rng() >> 2}};
        if (H::geq_p(x.l)) H::sub_p(x.l);
        if (!(H::mul_hi2(r, x) == H::mul(r, x)) || !(H::mul_challenge(r, x) == H::mul_sos(r, x)) || !(H::mul_challenge(x, r) == H::mul(x, r))) bad++;
    }
    printf("mismatches %zu\n", bad);
    H::Fr a = H::from_u64(123456789), b = H::from_u64(987654321);
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < 10000000; i++) a = H::mul(a, b);
    auto t1 = std::chrono::steady_clock::now();
    printf("mul %.2f ns (%llu)\n", std::chrono::duration<double, std::nano>(t1 - t0).count() / 1e7, (unsigned long long)a.l[0]);
    return bad != 0;
}
