/* Plain-C use of the boundary (what a cgo / Rust FFI caller does): upload two operands, run the degree-2
 * dot-product sumcheck with the transcript on the device, read the proof.
 *   gcc -std=c99 -I include examples/prove_dot.c -L jolt-atlas_amd -latlas_hip -Wl,-rpath,$PWD/jolt-atlas_amd -o prove_dot
 * Exits 0 and prints the transcript state on success; without a usable GPU it reports the library's error. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "atlas_hip.h"

#define CHECK(x) do { int rc_ = (x); if (rc_) { fprintf(stderr, "%s -> %d: %s\n", #x, rc_, atlas_last_error()); return 1; } } while (0)

int main(int argc, char **argv) {
    const size_t n_vars = argc > 1 ? (size_t)atoi(argv[1]) : 16, n = (size_t)1 << n_vars;
    CHECK(atlas_init(0));
    /* small Montgomery-form operands: i * R mod p is not needed for a demo — any limbs below p are field elements */
    atlas_fr_t *L = calloc(n, sizeof *L), *R = calloc(n, sizeof *R);
    for (size_t i = 0; i < n; i++) { L[i].l[0] = 3 * i + 1; R[i].l[0] = 5 * i + 2; }
    atlas_poly_t pl, pr;
    CHECK(atlas_poly_upload_fr(L, n, &pl));
    CHECK(atlas_poly_upload_fr(R, n, &pr));
    atlas_dot_prover_t prover;
    CHECK(atlas_dot_prover_new(pl, pr, NULL, ATLAS_EQ_NONE, 0, 0, &prover));
    atlas_fr_t claim;
    CHECK(atlas_dot_input_claim(prover, &claim));
    atlas_transcript_t t;
    CHECK(atlas_transcript_new(&t, (const uint8_t *)"example", 7));
    atlas_fr_t *proof = calloc(n_vars * 2, sizeof *proof), finals[3];
    atlas_u128_t *ch = calloc(n_vars, sizeof *ch);
    CHECK(atlas_sumcheck_prove_dot(prover, &claim, &t, proof, ch, finals));
    printf("rounds %zu, transcript state ", n_vars);
    for (int i = 0; i < 32; i++) printf("%02x", t.state[i]);
    printf("\n");
    CHECK(atlas_dot_prover_free(prover));
    free(L); free(R); free(proof); free(ch);
    return 0;
}
