/* C99 caller of the streaming host-to-host sessions (include/arkmpc.h, arkmpc_hostmul_*): what a Rust / cgo binding of a patched
 * AuthenticatedScalarResult::batch_mul sees (authenticated_scalar.rs:848-879 with Vec<ScalarShare> operands).  Two parties, host vectors with
 * nothing but malloc's alignment, payloads handed over in host memory; every word must equal the two-call host-buffer entry points
 * (arkmpc_beaver_mask + arkmpc_beaver_finish_fused) on the same inputs; sizes on both sides of the chunking threshold; the progress counter is
 * monotone; misuse is a status code and ends the session.  Exit 0 = all checks passed, 3 = no device (no CPU fallback). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "arkmpc.h"

#define CHECK(c) do { if (!(c)) { printf("FAILED line %d: %s (%s)\n", __LINE__, #c, arkmpc_last_error(ctx)); return 1; } } while (0)

static uint64_t rng_state = 0x13198A2E03707344ull;
static uint64_t rnd(void) { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }
static void fill(uint64_t* v, size_t elems) { for (size_t i = 0; i < elems; ++i) { for (int k = 0; k < 4; ++k) v[4 * i + k] = rnd(); v[4 * i + 3] &= 0x1fffffffffffffffull; } }

static int run(arkmpc_ctx* ctx, arkmpc_ctx* hctx, size_t n, int pinned) {
    uint64_t *buf[2][9], key[2][4];          /* x y a b c | de out | want_de want */
    for (int p = 0; p < 2; ++p) {
        for (int k = 0; k < 9; ++k) {
            void* q = NULL;
            if (pinned && k < 7) { CHECK(arkmpc_host_alloc(n * 64 + 16, &q) == ARKMPC_OK); }
            else q = malloc(n * 64 + 16);
            CHECK(q != NULL);
            buf[p][k] = (uint64_t*)q + 1;    /* 8 bytes off whatever alignment the allocator gave: a Vec promises no more */
            if (k < 5) fill(buf[p][k], 2 * n); else memset(buf[p][k], 0xA5, n * 64);
        }
        fill(key[p], 1);
    }
    for (int p = 0; p < 2; ++p) CHECK(arkmpc_beaver_mask(hctx, n, buf[p][0], buf[p][1], buf[p][2], buf[p][3], buf[p][7]) == ARKMPC_OK);
    for (int p = 0; p < 2; ++p)
        CHECK(arkmpc_beaver_finish_fused(hctx, n, p, key[p], buf[p][7], buf[1 - p][7], buf[p][2], buf[p][3], buf[p][4], buf[p][8]) == ARKMPC_OK);
    arkmpc_hostmul* s[2] = {NULL, NULL};
    for (int p = 0; p < 2; ++p) CHECK(arkmpc_hostmul_begin(ctx, n, buf[p][0], buf[p][1], buf[p][2], buf[p][3], buf[p][4], buf[p][5], &s[p]) == ARKMPC_OK && s[p]);
    for (int p = 0; p < 2; ++p) {
        size_t g0 = 0, g1 = 0;
        CHECK(arkmpc_hostmul_poll_de(s[p], &g0) == ARKMPC_OK && g0 <= n);
        CHECK(arkmpc_hostmul_wait_de(s[p]) == ARKMPC_OK);
        CHECK(arkmpc_hostmul_poll_de(s[p], &g1) == ARKMPC_OK && g1 == n && g1 >= g0);
        CHECK(memcmp(buf[p][5], buf[p][7], n * 64) == 0);                                    /* the payload this party sends */
    }
    for (int p = 0; p < 2; ++p) {
        CHECK(arkmpc_hostmul_finish(s[p], p, key[p], buf[1 - p][5], buf[p][6]) == ARKMPC_OK);
        CHECK(memcmp(buf[p][6], buf[p][8], n * 64) == 0);
    }
    /* the same gate with the payloads in their wire form: each party's frame must be the codec's own frame of its d||e (arkmpc_wire_encode_scalar_batch
     * on the host-buffer context), and the results the same words again */
    {
        size_t cap = 0, len[2] = {0, 0}, want_len = 0;
        CHECK(arkmpc_wire_frame_bound(2 * n, &cap) == ARKMPC_OK);
        uint8_t* fr[3];
        for (int k = 0; k < 3; ++k) { fr[k] = (uint8_t*)malloc(cap); CHECK(fr[k] != NULL); }
        for (int p = 0; p < 2; ++p) {
            memset(buf[p][6], 0x5A, n * 64);
            CHECK(arkmpc_hostmul_begin_wire(ctx, n, buf[p][0], buf[p][1], buf[p][2], buf[p][3], buf[p][4], 40 + (uint64_t)p, fr[p], cap, &len[p], &s[p]) == ARKMPC_OK && s[p]);
            CHECK(arkmpc_wire_encode_scalar_batch(hctx, 40 + (uint64_t)p, 2 * n, buf[p][7], fr[2], cap, &want_len) == ARKMPC_OK);
            CHECK(want_len == len[p] && memcmp(fr[p], fr[2], want_len) == 0);
        }
        for (int p = 0; p < 2; ++p) {
            uint64_t rid = 0;
            CHECK(arkmpc_hostmul_finish_wire(s[p], p, key[p], fr[1 - p], len[1 - p], buf[p][6], &rid) == ARKMPC_OK && rid == 40 + (uint64_t)(1 - p));
            CHECK(memcmp(buf[p][6], buf[p][8], n * 64) == 0);
        }
        /* a truncated peer frame is a status and ends the session */
        CHECK(arkmpc_hostmul_begin_wire(ctx, n, buf[0][0], buf[0][1], buf[0][2], buf[0][3], buf[0][4], 1, fr[2], cap, &want_len, &s[0]) == ARKMPC_OK);
        CHECK(arkmpc_hostmul_finish_wire(s[0], 0, key[0], fr[1], len[1] - 1, buf[0][6], NULL) == ARKMPC_ERR_BAD_ARG);
        for (int k = 0; k < 3; ++k) free(fr[k]);
    }
    for (int p = 0; p < 2; ++p)
        for (int k = 0; k < 9; ++k) { void* q = buf[p][k] - 1; if (pinned && k < 7) CHECK(arkmpc_host_free(q) == ARKMPC_OK); else free(q); }
    return 0;
}

int main(int argc, char** argv) {
    arkmpc_ctx *ctx = NULL, *hctx = NULL;
    int rc = arkmpc_ctx_create(ARKMPC_BN254_FR, 0, &ctx);
    if (rc == ARKMPC_ERR_NO_DEVICE) { printf("no device: status %d\n", rc); return 3; }
    if (rc != ARKMPC_OK || arkmpc_ctx_create(ARKMPC_BN254_FR, 0, &hctx) != ARKMPC_OK) return 1;
    CHECK(arkmpc_ctx_set_host_buffers(hctx, 1) == ARKMPC_OK);
    const size_t big = argc > 1 ? (size_t)strtoull(argv[1], NULL, 10) : 300007;
    const size_t sizes[] = {0, 1, 1000, 16384, 70001, big};
    for (size_t i = 0; i < sizeof sizes / sizeof sizes[0]; ++i)
        for (int pinned = 0; pinned < 2; ++pinned)
            if (sizes[i] == 0) {
                arkmpc_hostmul* s = NULL; size_t g = 7;
                CHECK(arkmpc_hostmul_begin(ctx, 0, NULL, NULL, NULL, NULL, NULL, NULL, &s) == ARKMPC_OK && s);
                CHECK(arkmpc_hostmul_poll_de(s, &g) == ARKMPC_OK && g == 0);
                uint64_t k[4] = {1, 0, 0, 0};
                CHECK(arkmpc_hostmul_finish(s, 0, k, NULL, NULL) == ARKMPC_OK);
            } else if (run(ctx, hctx, sizes[i], pinned)) {
                printf("size %zu pinned %d\n", sizes[i], pinned);
                return 1;
            }
    /* misuse: null operand = status, no session; bad party id in phase 2 = status AND the session is over; abort after phase 1 */
    {
        uint64_t v[8 * 4], de[8 * 4], out[8 * 4], k[4] = {1, 0, 0, 0};
        fill(v, 8);
        arkmpc_hostmul* s = NULL;
        CHECK(arkmpc_hostmul_begin(ctx, 4, NULL, v, v, v, v, de, &s) == ARKMPC_ERR_BAD_ARG && s == NULL);
        CHECK(arkmpc_hostmul_begin(ctx, 4, v, v, v, v, v, de, NULL) == ARKMPC_ERR_BAD_ARG);
        CHECK(arkmpc_hostmul_begin(ctx, 4, v, v, v, v, v, de, &s) == ARKMPC_OK);
        CHECK(arkmpc_hostmul_finish(s, 2, k, de, out) == ARKMPC_ERR_BAD_ARG);
        CHECK(arkmpc_hostmul_begin(ctx, 4, v, v, v, v, v, de, &s) == ARKMPC_OK);
        CHECK(arkmpc_hostmul_abort(s) == ARKMPC_OK);
        CHECK(arkmpc_hostmul_poll_de(NULL, NULL) == ARKMPC_ERR_BAD_ARG && arkmpc_hostmul_wait_de(NULL) == ARKMPC_ERR_BAD_ARG);
        CHECK(arkmpc_hostmul_finish(NULL, 0, k, de, out) == ARKMPC_ERR_BAD_ARG && arkmpc_hostmul_abort(NULL) == ARKMPC_ERR_BAD_ARG);
        CHECK(arkmpc_host_register(NULL, 16) == ARKMPC_ERR_BAD_ARG && arkmpc_host_alloc(16, NULL) == ARKMPC_ERR_BAD_ARG && arkmpc_host_free(NULL) == ARKMPC_OK);
    }
    arkmpc_ctx_destroy(hctx);
    arkmpc_ctx_destroy(ctx);
    printf("hostmul sessions ok\n");
    return 0;
}
