// Replays the reference-held unit vectors of tests/golden/ref_unit_vectors.json (passed as integers on the command line by
// tests/test_ref_unit_vectors.py) through the PRODUCT's host-side univariate helpers (csrc/host_field.hpp, csrc/host_poly.hpp):
//   from_evals_and_hint <deg> <hint> <e0> <e2> [<e3>]   -> coefficients      (UniPoly::from_evals_and_hint, unipoly.rs:66-98)
//   toom <e0> .. <e_{n-2}> <e_inf>                      -> coefficients      (UniPoly::from_evals_toom, unipoly.rs:103-134)
//   gauss <n> <row-major n x (n+1)>                     -> solution          (utils/gaussian_elimination.rs)
//   trim <c0> ..                                        -> coefficients      (UniPoly::from_coeff, unipoly.rs:39-52)
// Prints the canonical integers (all vectors are small).   g++ -O2 -std=c++17 -I. tools/check_host_poly.cpp
#include <chrono>
#include <cstdio>
#include <random>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "jolt-atlas_amd/csrc/host_poly.hpp"
namespace H = atlas_host;
static void print(const std::vector<H::Fr>& v) {
    for (const H::Fr& x : v) {
        uint64_t c[4];
        H::to_canonical(x, c);
        if (c[1] | c[2] | c[3]) { printf("big "); continue; }
        printf("%llu ", (unsigned long long)c[0]);
    }
    printf("\n");
}
int main(int argc, char** argv) {
    if (argc < 2) return 2;
    std::vector<H::Fr> a;
    for (int i = 2; i < argc; i++) a.push_back(H::from_u64(strtoull(argv[i], nullptr, 10)));
    if (!strcmp(argv[1], "from_evals_and_hint")) {
        const int deg = (int)strtoull(argv[2], nullptr, 10);
        std::vector<H::Fr> c(4);
        const int n = H::unipoly_from_evals_and_hint(a[1], a.data() + 2, deg, c.data());
        c.resize(n);
        print(c);
    } else if (!strcmp(argv[1], "toom")) {
        print(H::from_evals_toom(a));
    } else if (!strcmp(argv[1], "toom_selfcheck")) {
        // the O(n) interpolation against the matrix form on random evaluations, every length the provers use and beyond; mul_small
        // against the full multiplication
        std::mt19937_64 rng(7);
        auto rnd = [&] { H::Fr x{{rng(), rng(), rng(), rng() >> 3}}; if (H::geq_p(x.l)) H::sub_p(x.l); return x; };
        size_t bad = 0;
        for (size_t n = 2; n <= 40; n++)
            for (int rep = 0; rep < 20; rep++) {
                std::vector<H::Fr> e(n);
                for (auto& x : e) x = rep == 0 ? H::zero() : rep == 1 ? H::sub(H::zero(), H::one()) : rnd();
                const auto u = H::from_evals_toom(e), v = H::from_evals_toom_matrix(e);
                for (size_t i = 0; i < n; i++) if (!(u[i] == v[i])) bad++;
            }
        for (int rep = 0; rep < 200000; rep++) {
            H::Fr x = rep < 64 ? H::sub(H::zero(), H::from_u64(rep)) : rnd();
            const uint32_t k = (uint32_t)(rng() & 63);
            if (!(H::mul_small(x, k) == H::mul(x, H::from_u64(k)))) bad++;
        }
        printf("%zu\n", bad);
        auto t0 = std::chrono::steady_clock::now();
        std::vector<H::Fr> e(18); for (auto& x : e) x = rnd();
        H::Fr acc = H::zero();
        for (int i = 0; i < 20000; i++) { e[3] = acc; acc = H::add(acc, H::from_evals_toom(e)[5]); }
        auto t1 = std::chrono::steady_clock::now();
        for (int i = 0; i < 20000; i++) { e[3] = acc; acc = H::add(acc, H::from_evals_toom_matrix(e)[5]); }
        auto t2 = std::chrono::steady_clock::now();
        fprintf(stderr, "degree-17 interpolation: %.2f us (matrix form %.2f us) %llu\n", std::chrono::duration<double, std::micro>(t1 - t0).count() / 20000,
                std::chrono::duration<double, std::micro>(t2 - t1).count() / 20000, (unsigned long long)acc.l[0]);
        return bad != 0;
    } else if (!strcmp(argv[1], "gauss")) {
        const size_t n = strtoull(argv[2], nullptr, 10);
        std::vector<std::vector<H::Fr>> m(n, std::vector<H::Fr>(n + 1));
        for (size_t i = 0; i < n; i++) for (size_t j = 0; j <= n; j++) m[i][j] = a[1 + i * (n + 1) + j];
        print(H::gauss_solve(m));
    } else if (!strcmp(argv[1], "trim")) {
        H::trim(a);
        print(a);
    } else return 2;
    return 0;
}
