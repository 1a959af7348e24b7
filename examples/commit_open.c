/* Plain-C use of the PCS seam: SRS for a known trapdoor, commit to a polynomial, open it at a point.
 *   gcc -std=c99 -I include examples/commit_open.c -L jolt-atlas_amd -latlas_hip -Wl,-rpath,$PWD/jolt-atlas_amd -o commit_open
 * (A production caller uploads the ceremony SRS with atlas_srs_upload instead of generating one.) */
#include <stdio.h>
#include <stdlib.h>
#include "atlas_hip.h"

#define CHECK(x) do { int rc_ = (x); if (rc_) { fprintf(stderr, "%s -> %d: %s\n", #x, rc_, atlas_last_error()); return 1; } } while (0)

int main(int argc, char **argv) {
    const size_t ell = argc > 1 ? (size_t)atoi(argv[1]) : 16, n = (size_t)1 << ell;
    CHECK(atlas_init(0));
    atlas_fr_t tau = {{0x1234567, 0, 0, 0}};                    /* any field element (Montgomery limbs) */
    atlas_srs_t srs;
    CHECK(atlas_srs_generate(&tau, n, &srs));
    atlas_fr_t *coeffs = calloc(n, sizeof *coeffs);
    for (size_t i = 0; i < n; i++) coeffs[i].l[0] = 7 * i + 3;
    atlas_poly_t poly;
    CHECK(atlas_poly_upload_fr(coeffs, n, &poly));
    atlas_g1_affine_t commitment;
    CHECK(atlas_msm_poly(srs, 0, poly, &commitment));            /* HyperKZG::commit */
    atlas_u128_t *point = calloc(ell, sizeof *point);
    for (size_t i = 0; i < ell; i++) { point[i].lo = 0x9e3779b97f4a7c15ull * (i + 1); point[i].hi = i + 1; }
    atlas_transcript_t t;
    CHECK(atlas_transcript_new(&t, (const uint8_t *)"example", 7));
    atlas_g1_affine_t *com = calloc(ell, sizeof *com), w[3];
    atlas_fr_t *v = calloc(3 * ell, sizeof *v);
    CHECK(atlas_hyperkzg_open(srs, poly, point, ell, &t, com, w, v));
    printf("ell %zu: commitment.x[0] = %016llx, %zu intermediate commitments, transcript state ", ell,
           (unsigned long long)commitment.x.l[0], ell - 1);
    for (int i = 0; i < 32; i++) printf("%02x", t.state[i]);
    printf("\n");
    CHECK(atlas_poly_free(poly));
    CHECK(atlas_srs_free(srs));
    free(coeffs); free(point); free(com); free(v);
    return 0;
}
