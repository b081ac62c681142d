/* C99 caller of the device-batch carrier (include/arkmpc.h, arkmpc_batch_*): the handle a ResultValue::DeviceBatch variant would
 * hold (fabric/result.rs:47-64).  A two-party Beaver multiplication runs on handles in both layouts and must equal, word for
 * word, the pointer-level host-buffer entry points on the same inputs; slices share storage and outlive their parent handle;
 * misuse is a status code.  Exit 0 = all checks passed, 3 = no device (no CPU fallback). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "arkmpc.h"

#define N 1000
#define CHECK(c) do { if (!(c)) { printf("FAILED line %d: %s (%s)\n", __LINE__, #c, arkmpc_last_error(ctx)); return 1; } } while (0)

static uint64_t rng_state = 0x243F6A8885A308D3ull;
static uint64_t rnd(void) { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }
/* n field elements below 2^253 (< every modulus here): valid Montgomery residues */
static void fill(uint64_t* v, size_t elems) { for (size_t i = 0; i < elems; ++i) { for (int k = 0; k < 4; ++k) v[4 * i + k] = rnd(); v[4 * i + 3] &= 0x1fffffffffffffffull; } }

int main(void) {
    arkmpc_ctx *ctx = NULL, *hctx = NULL;
    int rc = arkmpc_ctx_create(ARKMPC_BN254_FR, 0, &ctx);
    if (rc == ARKMPC_ERR_NO_DEVICE) { printf("no device: status %d\n", rc); return 3; }
    if (rc != ARKMPC_OK || arkmpc_ctx_create(ARKMPC_BN254_FR, 0, &hctx) != ARKMPC_OK) return 1;
    CHECK(arkmpc_ctx_set_host_buffers(hctx, 1) == ARKMPC_OK);
    static uint64_t x[2][8 * N], y[2][8 * N], a[2][8 * N], b[2][8 * N], c[2][8 * N], key[2][4];
    static uint64_t want_de[2][8 * N], want[2][8 * N], got_de[8 * N], got[8 * N];
    for (int p = 0; p < 2; ++p) { fill(x[p], 2 * N); fill(y[p], 2 * N); fill(a[p], 2 * N); fill(b[p], 2 * N); fill(c[p], 2 * N); fill(key[p], 1); }
    /* expectation: the pointer-level entry points in host-buffer mode */
    for (int p = 0; p < 2; ++p) CHECK(arkmpc_beaver_mask(hctx, N, x[p], y[p], a[p], b[p], want_de[p]) == ARKMPC_OK);
    for (int p = 0; p < 2; ++p) CHECK(arkmpc_beaver_finish_fused(hctx, N, p, key[p], want_de[p], want_de[1 - p], a[p], b[p], c[p], want[p]) == ARKMPC_OK);
    for (int layout = ARKMPC_LAYOUT_AOS; layout <= ARKMPC_LAYOUT_SPLIT; ++layout) {
        arkmpc_batch *bx[2], *by[2], *ba[2], *bb[2], *bc[2], *de[2], *out[2];
        for (int p = 0; p < 2; ++p) {
            CHECK(arkmpc_batch_from_host(ctx, ARKMPC_KIND_SCALAR_SHARE, layout, N, x[p], &bx[p]) == ARKMPC_OK);
            CHECK(arkmpc_batch_from_host(ctx, ARKMPC_KIND_SCALAR_SHARE, layout, N, y[p], &by[p]) == ARKMPC_OK);
            CHECK(arkmpc_batch_from_host(ctx, ARKMPC_KIND_SCALAR_SHARE, layout, N, a[p], &ba[p]) == ARKMPC_OK);
            CHECK(arkmpc_batch_from_host(ctx, ARKMPC_KIND_SCALAR_SHARE, 1 - layout, N, b[p], &bb[p]) == ARKMPC_OK);     /* mixed operand layouts */
            CHECK(arkmpc_batch_from_host(ctx, ARKMPC_KIND_SCALAR_SHARE, layout, N, c[p], &bc[p]) == ARKMPC_OK);
            CHECK(arkmpc_batch_len(bx[p]) == N && arkmpc_batch_kind(bx[p]) == ARKMPC_KIND_SCALAR_SHARE && arkmpc_batch_layout(bx[p]) == layout);
            CHECK(arkmpc_batch_elem_words(bx[p]) == 8 && arkmpc_batch_stride(bx[p]) == (layout == ARKMPC_LAYOUT_AOS ? 8u : 4u));
            CHECK(arkmpc_batch_beaver_mask(ctx, bx[p], by[p], ba[p], bb[p], &de[p]) == ARKMPC_OK);
            CHECK(arkmpc_batch_kind(de[p]) == ARKMPC_KIND_SCALAR && arkmpc_batch_len(de[p]) == 2 * N);
            CHECK(arkmpc_batch_to_host(ctx, de[p], got_de) == ARKMPC_OK);
            CHECK(memcmp(got_de, want_de[p], sizeof got_de) == 0);
        }
        for (int p = 0; p < 2; ++p) {
            CHECK(arkmpc_batch_beaver_finish(ctx, p, key[p], de[p], de[1 - p], ba[p], bb[p], bc[p], layout, &out[p]) == ARKMPC_OK);
            CHECK(arkmpc_batch_layout(out[p]) == layout && arkmpc_batch_len(out[p]) == N);
            CHECK(arkmpc_batch_to_host(ctx, out[p], got) == ARKMPC_OK);                 /* always arkworks records */
            CHECK(memcmp(got, want[p], sizeof got) == 0);
        }
        /* slices: &out[100..350] shares the storage, survives the parent handle, and a slice of it re-bases correctly */
        arkmpc_batch *s1 = NULL, *s2 = NULL;
        CHECK(arkmpc_batch_slice(ctx, out[0], 100, 250, &s1) == ARKMPC_OK && arkmpc_batch_len(s1) == 250);
        CHECK(arkmpc_batch_slice(ctx, out[0], N - 1, 2, &s2) == ARKMPC_ERR_BAD_ARG && s2 == NULL);     /* out of range */
        CHECK(arkmpc_batch_destroy(ctx, out[0]) == ARKMPC_OK);                          /* parent handle dropped first */
        CHECK(arkmpc_batch_to_host(ctx, s1, got) == ARKMPC_OK && memcmp(got, want[0] + 8 * 100, 250 * 64) == 0);
        CHECK(arkmpc_batch_slice(ctx, s1, 50, 10, &s2) == ARKMPC_OK);
        CHECK(arkmpc_batch_to_host(ctx, s2, got) == ARKMPC_OK && memcmp(got, want[0] + 8 * 150, 10 * 64) == 0);
        /* a gate on a sub-range: the last 250 gates through slices of every operand == the same rows of the full result */
        {
            arkmpc_batch *sa, *sb, *sc, *sd0, *sd1, *se0, *se1, *r = NULL;
            CHECK(arkmpc_batch_slice(ctx, ba[1], 750, 250, &sa) == ARKMPC_OK && arkmpc_batch_slice(ctx, bb[1], 750, 250, &sb) == ARKMPC_OK);
            CHECK(arkmpc_batch_slice(ctx, bc[1], 750, 250, &sc) == ARKMPC_OK);
            /* d||e of a sub-range is not contiguous in the full payload, so rebuild it from the d and e slices */
            CHECK(arkmpc_batch_slice(ctx, de[1], 750, 250, &sd1) == ARKMPC_OK && arkmpc_batch_slice(ctx, de[1], N + 750, 250, &se1) == ARKMPC_OK);
            CHECK(arkmpc_batch_slice(ctx, de[0], 750, 250, &sd0) == ARKMPC_OK && arkmpc_batch_slice(ctx, de[0], N + 750, 250, &se0) == ARKMPC_OK);
            arkmpc_batch *m = NULL, *q = NULL;
            CHECK(arkmpc_batch_create(ctx, ARKMPC_KIND_SCALAR, ARKMPC_LAYOUT_AOS, 500, &m) == ARKMPC_OK && arkmpc_batch_create(ctx, ARKMPC_KIND_SCALAR, ARKMPC_LAYOUT_AOS, 500, &q) == ARKMPC_OK);
            CHECK(arkmpc_memcpy_d2d(ctx, arkmpc_batch_data(m), arkmpc_batch_data(sd1), 250 * 32) == ARKMPC_OK);
            CHECK(arkmpc_memcpy_d2d(ctx, arkmpc_batch_data(m) + 4 * 250, arkmpc_batch_data(se1), 250 * 32) == ARKMPC_OK);
            CHECK(arkmpc_memcpy_d2d(ctx, arkmpc_batch_data(q), arkmpc_batch_data(sd0), 250 * 32) == ARKMPC_OK);
            CHECK(arkmpc_memcpy_d2d(ctx, arkmpc_batch_data(q) + 4 * 250, arkmpc_batch_data(se0), 250 * 32) == ARKMPC_OK);
            CHECK(arkmpc_batch_beaver_finish(ctx, 1, key[1], m, q, sa, sb, sc, ARKMPC_LAYOUT_AOS, &r) == ARKMPC_OK);
            CHECK(arkmpc_batch_to_host(ctx, r, got) == ARKMPC_OK && memcmp(got, want[1] + 8 * 750, 250 * 64) == 0);
            /* misuse is a status, never a crash or an out-of-bounds read: short peer payload, wrong kind, foreign context */
            arkmpc_batch* bad = NULL;
            CHECK(arkmpc_batch_beaver_finish(ctx, 1, key[1], m, sd0, sa, sb, sc, ARKMPC_LAYOUT_AOS, &bad) == ARKMPC_ERR_BAD_ARG && bad == NULL);
            CHECK(arkmpc_batch_beaver_finish(ctx, 1, key[1], m, q, sa, sb, m, ARKMPC_LAYOUT_AOS, &bad) == ARKMPC_ERR_BAD_ARG);
            CHECK(arkmpc_batch_beaver_mask(ctx, sa, sb, sc, bx[0], &bad) == ARKMPC_ERR_BAD_ARG);
            CHECK(arkmpc_batch_create(ctx, ARKMPC_KIND_POINT, ARKMPC_LAYOUT_SPLIT, 4, &bad) == ARKMPC_ERR_BAD_ARG);
            CHECK(arkmpc_batch_create(hctx, ARKMPC_KIND_SCALAR, ARKMPC_LAYOUT_AOS, 4, &bad) == ARKMPC_ERR_BAD_ARG);
            arkmpc_batch* pts = NULL;
            CHECK(arkmpc_batch_create(ctx, ARKMPC_KIND_POINT_SHARE, ARKMPC_LAYOUT_AOS, 3, &pts) == ARKMPC_OK && arkmpc_batch_elem_words(pts) == 24);
            arkmpc_batch* all[] = {sa, sb, sc, sd0, sd1, se0, se1, m, q, r, pts};
            for (size_t i = 0; i < sizeof all / sizeof all[0]; ++i) CHECK(arkmpc_batch_destroy(ctx, all[i]) == ARKMPC_OK);
        }
        CHECK(arkmpc_batch_destroy(ctx, s1) == ARKMPC_OK && arkmpc_batch_destroy(ctx, s2) == ARKMPC_OK);
        CHECK(arkmpc_batch_destroy(ctx, out[1]) == ARKMPC_OK);
        for (int p = 0; p < 2; ++p) {
            arkmpc_batch* all[] = {bx[p], by[p], ba[p], bb[p], bc[p], de[p]};
            for (size_t i = 0; i < 6; ++i) CHECK(arkmpc_batch_destroy(ctx, all[i]) == ARKMPC_OK);
        }
    }
    /* empty batches are legal values (authenticated_scalar.rs:853-855) */
    {
        arkmpc_batch *e0 = NULL, *de0 = NULL;
        CHECK(arkmpc_batch_create(ctx, ARKMPC_KIND_SCALAR_SHARE, ARKMPC_LAYOUT_AOS, 0, &e0) == ARKMPC_OK && arkmpc_batch_len(e0) == 0);
        CHECK(arkmpc_batch_beaver_mask(ctx, e0, e0, e0, e0, &de0) == ARKMPC_OK && arkmpc_batch_len(de0) == 0);
        CHECK(arkmpc_batch_to_host(ctx, de0, NULL) == ARKMPC_OK);
        CHECK(arkmpc_batch_destroy(ctx, de0) == ARKMPC_OK && arkmpc_batch_destroy(ctx, e0) == ARKMPC_OK);
    }
    arkmpc_ctx_destroy(hctx);
    arkmpc_ctx_destroy(ctx);
    printf("batch carrier ok\n");
    return 0;
}
