/* C99 caller of the multi-device group (include/arkmpc.h, arkmpc_group_*): ONE process, G members.  Run with repeated device ids
 * ({0,0,...}: members share the GPU, each with its own stream) the whole sharded path -- range kernels, peer pushes, gathers, the
 * pipelined commitment, the AND-reduced verify flag -- must equal, word for word, what ONE context computes on the unsharded batch:
 *   config-3 shape: two parties' Beaver batch_mul over n gates of BN254 Fr, in both layouts, d||e handed over member by member; the same as
 *                   streaming sessions over the group (arkmpc_group_hostmul_*: host vectors in and out, one range session per member);
 *   config-5 shape: open_authenticated_batch over n shares of BLS12-381 Fr incl. commitments, and a corrupted last share;
 *   MSM: group bucket MSM == single-context MSM (as affine points).
 * usage: group_oversub [n [G [device ids...]]]     (default n = 100003, G = 4 on device 0; device ids default to 0)
 * Exit 0 = all checks passed, 3 = no device (no CPU fallback). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "arkmpc.h"

#define MAXG 16
#define CHECK(c) do { if (!(c)) { printf("FAILED line %d: %s (ctx: %s | group: %s)\n", __LINE__, #c, arkmpc_last_error(ctx), grp[0] ? arkmpc_group_last_error(grp[0]) : ""); return 1; } } while (0)

static uint64_t rng_state = 0x243F6A8885A308D3ull;
static uint64_t rnd(void) { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }
/* field elements below 2^252 (< every modulus here): valid Montgomery residues */
static uint64_t* fill(size_t elems) {
    uint64_t* v = (uint64_t*)malloc((elems ? elems : 1) * 32);
    if (!v) { printf("out of host memory\n"); exit(1); }
    for (size_t i = 0; i < elems; ++i) { for (int k = 0; k < 4; ++k) v[4 * i + k] = rnd(); v[4 * i + 3] &= 0x0fffffffffffffffull; }
    return v;
}
static uint64_t* zeros(size_t words) { uint64_t* v = (uint64_t*)calloc(words ? words : 1, 8); if (!v) { printf("out of host memory\n"); exit(1); } return v; }
typedef uint64_t* shards_t[MAXG];
#define CS(x) ((const uint64_t* const*)(x))

static int beaver(size_t n, int G, const int* devs) {
    arkmpc_ctx* ctx = NULL;
    arkmpc_group* grp[2] = {NULL, NULL};
    CHECK(arkmpc_ctx_create(ARKMPC_BN254_FR, devs[0], &ctx) == ARKMPC_OK);
    for (int p = 0; p < 2; ++p) CHECK(arkmpc_group_create(ARKMPC_BN254_FR, G, devs, &grp[p]) == ARKMPC_OK);
    CHECK(arkmpc_group_size(grp[0]) == G && arkmpc_group_device(grp[0], G - 1) == devs[G - 1] && arkmpc_group_ctx(grp[0], 0) != NULL);
    size_t covered = 0;
    for (int m = 0; m < G; ++m) {                                  /* ranges tile [0, n) in order */
        size_t lo, cnt;
        CHECK(arkmpc_group_shard_range(grp[0], n, m, &lo, &cnt) == ARKMPC_OK && lo == covered && lo == (n * (size_t)m) / (size_t)G);
        covered += cnt;
        for (int k = 0; k < G; ++k) CHECK(arkmpc_group_peer_access(grp[0], m, k) == 1 || devs[m] != devs[k]);
    }
    CHECK(covered == n);
    uint64_t *x[2], *y[2], *a[2], *b[2], *c[2], *key[2], *want_de[2], *want[2];
    uint64_t* got = zeros(8 * n);
    for (int p = 0; p < 2; ++p) { x[p] = fill(2 * n); y[p] = fill(2 * n); a[p] = fill(2 * n); b[p] = fill(2 * n); c[p] = fill(2 * n); key[p] = fill(1); want_de[p] = zeros(8 * n); want[p] = zeros(8 * n); }
    /* expectation: ONE context, the whole batch */
    {
        void *dx, *dy, *da, *db, *dc, *dde[2], *dout;
        CHECK(arkmpc_malloc(ctx, n * 64 + 16, &dx) == 0 && arkmpc_malloc(ctx, n * 64 + 16, &dy) == 0 && arkmpc_malloc(ctx, n * 64 + 16, &da) == 0 &&
              arkmpc_malloc(ctx, n * 64 + 16, &db) == 0 && arkmpc_malloc(ctx, n * 64 + 16, &dc) == 0 && arkmpc_malloc(ctx, n * 64 + 16, &dout) == 0 &&
              arkmpc_malloc(ctx, n * 64 + 16, &dde[0]) == 0 && arkmpc_malloc(ctx, n * 64 + 16, &dde[1]) == 0);
        for (int p = 0; p < 2; ++p) {
            CHECK(arkmpc_memcpy_h2d(ctx, dx, x[p], n * 64) == 0 && arkmpc_memcpy_h2d(ctx, dy, y[p], n * 64) == 0 && arkmpc_memcpy_h2d(ctx, da, a[p], n * 64) == 0 &&
                  arkmpc_memcpy_h2d(ctx, db, b[p], n * 64) == 0);
            CHECK(arkmpc_beaver_mask(ctx, n, dx, dy, da, db, dde[p]) == ARKMPC_OK);
            CHECK(arkmpc_memcpy_d2h(ctx, want_de[p], dde[p], n * 64) == 0);
        }
        for (int p = 0; p < 2; ++p) {
            CHECK(arkmpc_memcpy_h2d(ctx, da, a[p], n * 64) == 0 && arkmpc_memcpy_h2d(ctx, db, b[p], n * 64) == 0 && arkmpc_memcpy_h2d(ctx, dc, c[p], n * 64) == 0);
            CHECK(arkmpc_beaver_finish_fused(ctx, n, p, key[p], dde[p], dde[1 - p], da, db, dc, dout) == ARKMPC_OK);
            CHECK(arkmpc_memcpy_d2h(ctx, want[p], dout, n * 64) == 0);
        }
        arkmpc_free(ctx, dx); arkmpc_free(ctx, dy); arkmpc_free(ctx, da); arkmpc_free(ctx, db); arkmpc_free(ctx, dc); arkmpc_free(ctx, dout); arkmpc_free(ctx, dde[0]); arkmpc_free(ctx, dde[1]);
    }
    for (int layout = ARKMPC_LAYOUT_AOS; layout <= ARKMPC_LAYOUT_SPLIT; ++layout) {
        const size_t segs = layout == ARKMPC_LAYOUT_SPLIT ? 2 : 1, ew = layout == ARKMPC_LAYOUT_SPLIT ? 4 : 8;
        shards_t sx[2], sy[2], sa[2], sb[2], sc[2], sde[2], sout[2];
        for (int p = 0; p < 2; ++p) {
            CHECK(arkmpc_group_malloc(grp[p], n, segs, ew, sx[p]) == 0 && arkmpc_group_malloc(grp[p], n, segs, ew, sy[p]) == 0 &&
                  arkmpc_group_malloc(grp[p], n, segs, ew, sa[p]) == 0 && arkmpc_group_malloc(grp[p], n, segs, ew, sb[p]) == 0 &&
                  arkmpc_group_malloc(grp[p], n, segs, ew, sc[p]) == 0 && arkmpc_group_malloc(grp[p], n, segs, ew, sout[p]) == 0 &&
                  arkmpc_group_malloc(grp[p], n, 2, 4, sde[p]) == 0);
            CHECK(arkmpc_group_shares_from_host(grp[p], layout, n, x[p], sx[p]) == 0 && arkmpc_group_shares_from_host(grp[p], layout, n, y[p], sy[p]) == 0 &&
                  arkmpc_group_shares_from_host(grp[p], layout, n, a[p], sa[p]) == 0 && arkmpc_group_shares_from_host(grp[p], layout, n, b[p], sb[p]) == 0 &&
                  arkmpc_group_shares_from_host(grp[p], layout, n, c[p], sc[p]) == 0);
            CHECK(arkmpc_group_beaver_mask(grp[p], layout, n, CS(sx[p]), CS(sy[p]), CS(sa[p]), CS(sb[p]), sde[p]) == ARKMPC_OK);
            /* the payload a party sends: shards -> host, each member over its own link; == the unsharded d||e */
            CHECK(arkmpc_group_gather_d2h(grp[p], n, 2, 4, CS(sde[p]), got) == ARKMPC_OK);
            CHECK(memcmp(got, want_de[p], n * 64) == 0);
        }
        /* both parties' K1 have drained (gather_d2h blocks), so the member-by-member hand-over needs no further ordering */
        for (int p = 0; p < 2; ++p) {
            CHECK(arkmpc_group_beaver_finish_fused(grp[p], layout, n, p, key[p], CS(sde[p]), CS(sde[1 - p]), CS(sa[p]), CS(sb[p]), CS(sc[p]), sout[p]) == ARKMPC_OK);
            CHECK(arkmpc_group_shares_to_host(grp[p], layout, n, CS(sout[p]), got) == ARKMPC_OK);
            CHECK(memcmp(got, want[p], n * 64) == 0);
        }
        /* the peer's payload arriving from the network: host -> shards, then K2+K3 again into fresh outputs */
        {
            shards_t speer;
            CHECK(arkmpc_group_malloc(grp[0], n, 2, 4, speer) == 0);
            CHECK(arkmpc_group_scatter_h2d(grp[0], n, 2, 4, want_de[1], speer) == ARKMPC_OK);
            CHECK(arkmpc_group_beaver_finish_fused(grp[0], layout, n, 0, key[0], CS(sde[0]), CS(speer), CS(sa[0]), CS(sb[0]), CS(sc[0]), sout[0]) == ARKMPC_OK);
            CHECK(arkmpc_group_shares_to_host(grp[0], layout, n, CS(sout[0]), got) == ARKMPC_OK && memcmp(got, want[0], n * 64) == 0);
            CHECK(arkmpc_group_free(grp[0], speer) == 0);
        }
        /* device gathers as peer writes: gather to the last member, K1 storing straight into the root buffer, all-gather, scatter */
        {
            const int root = G - 1;
            arkmpc_ctx* rctx = arkmpc_group_ctx(grp[0], root);
            void *full = NULL, *full2 = NULL;
            CHECK(arkmpc_malloc(rctx, n * 64 + 16, &full) == 0 && arkmpc_malloc(rctx, n * 64 + 16, &full2) == 0);
            CHECK(arkmpc_group_gather(grp[0], n, 2, 4, CS(sde[0]), root, (uint64_t*)full) == ARKMPC_OK);
            CHECK(arkmpc_memcpy_d2h(rctx, got, full, n * 64) == 0 && memcmp(got, want_de[0], n * 64) == 0);      /* on root's stream: ordered behind the pushes */
            CHECK(arkmpc_group_beaver_mask_gathered(grp[0], layout, n, CS(sx[0]), CS(sy[0]), CS(sa[0]), CS(sb[0]), root, (uint64_t*)full2) == ARKMPC_OK);
            CHECK(arkmpc_memcpy_d2h(rctx, got, full2, n * 64) == 0 && memcmp(got, want_de[0], n * 64) == 0);
            shards_t outs, back;
            for (int m = 0; m < G; ++m) { void* q; CHECK(arkmpc_malloc(arkmpc_group_ctx(grp[0], m), n * 64 + 16, &q) == 0); outs[m] = (uint64_t*)q; }
            CHECK(arkmpc_group_allgather(grp[0], n, 2, 4, CS(sde[0]), outs) == ARKMPC_OK);
            for (int m = 0; m < G; ++m) CHECK(arkmpc_memcpy_d2h(arkmpc_group_ctx(grp[0], m), got, outs[m], n * 64) == 0 && memcmp(got, want_de[0], n * 64) == 0);
            CHECK(arkmpc_group_malloc(grp[0], n, 2, 4, back) == 0);
            CHECK(arkmpc_group_scatter(grp[0], n, 2, 4, (const uint64_t*)full, root, back) == ARKMPC_OK);
            CHECK(arkmpc_group_gather_d2h(grp[0], n, 2, 4, CS(back), got) == ARKMPC_OK && memcmp(got, want_de[0], n * 64) == 0);
            for (int m = 0; m < G; ++m) arkmpc_free(arkmpc_group_ctx(grp[0], m), outs[m]);
            CHECK(arkmpc_group_free(grp[0], back) == 0);
            arkmpc_free(rctx, full); arkmpc_free(rctx, full2);
        }
        for (int p = 0; p < 2; ++p) {
            CHECK(arkmpc_group_free(grp[p], sx[p]) == 0 && arkmpc_group_free(grp[p], sy[p]) == 0 && arkmpc_group_free(grp[p], sa[p]) == 0 && arkmpc_group_free(grp[p], sb[p]) == 0 &&
                  arkmpc_group_free(grp[p], sc[p]) == 0 && arkmpc_group_free(grp[p], sout[p]) == 0 && arkmpc_group_free(grp[p], sde[p]) == 0);
        }
    }
    /* streaming sessions over the group: the same host vectors in, host vectors out, each member on its range and link.  Once with the
     * vectors as malloc gave them (pinned by the call), once in memory the caller pinned (arkmpc_host_alloc: the phases run in place). */
    for (int pinned = 0; pinned < 2; ++pinned) {
        uint64_t *hx[2], *hy[2], *ha[2], *hb[2], *hc[2], *hde[2], *hout[2];
        arkmpc_group_hostmul* ses[2] = {NULL, NULL};
        for (int p = 0; p < 2; ++p) {
            uint64_t** dst[7] = {&hx[p], &hy[p], &ha[p], &hb[p], &hc[p], &hde[p], &hout[p]};
            const uint64_t* src[7] = {x[p], y[p], a[p], b[p], c[p], NULL, NULL};
            for (int k = 0; k < 7; ++k) {
                if (pinned) { void* q = NULL; CHECK(arkmpc_host_alloc(n * 64 + 16, &q) == ARKMPC_OK); *dst[k] = (uint64_t*)q; }
                else *dst[k] = zeros(8 * n + 2);
                if (src[k]) memcpy(*dst[k], src[k], n * 64); else memset(*dst[k], 0, n * 64);
            }
        }
        for (int p = 0; p < 2; ++p) CHECK(arkmpc_group_hostmul_begin(grp[p], n, hx[p], hy[p], ha[p], hb[p], hc[p], hde[p], &ses[p]) == ARKMPC_OK);
        for (int p = 0; p < 2; ++p) {
            size_t done = 0;
            CHECK(arkmpc_group_hostmul_poll_de(ses[p], &done) == ARKMPC_OK && done <= n);
            CHECK(arkmpc_group_hostmul_wait_de(ses[p]) == ARKMPC_OK);
            CHECK(arkmpc_group_hostmul_poll_de(ses[p], &done) == ARKMPC_OK && done == n);
            CHECK(memcmp(hde[p], want_de[p], n * 64) == 0);
        }
        for (int p = 0; p < 2; ++p) {
            CHECK(arkmpc_group_hostmul_finish(ses[p], p, key[p], hde[1 - p], hout[p]) == ARKMPC_OK);
            CHECK(memcmp(hout[p], want[p], n * 64) == 0);
        }
        if (n >= (size_t)4096 * (size_t)G && pinned) {        /* every member's range reaches the in-place threshold: both phases of both parties ran as kernels on the vectors */
            for (int m = 0; m < G; ++m) {
                arkmpc_ctx_stats st;
                CHECK(arkmpc_ctx_get_stats(arkmpc_group_ctx(grp[0], m), &st) == ARKMPC_OK);
                CHECK(st.hostmul_zero_copy_phases[0] >= 1 && st.hostmul_zero_copy_phases[1] >= 1 && st.hostmul_device_bytes_peak <= 512 * (n / (size_t)G + 1));
            }
        }
        /* misuse: a null vector, a bad party; an aborted session leaves the group usable */
        {
            arkmpc_group_hostmul* s2 = NULL;
            if (n) CHECK(arkmpc_group_hostmul_begin(grp[0], n, hx[0], NULL, ha[0], hb[0], hc[0], hde[0], &s2) == ARKMPC_ERR_BAD_ARG && s2 == NULL);
            CHECK(arkmpc_group_hostmul_begin(grp[0], n, hx[0], hy[0], ha[0], hb[0], hc[0], hde[0], &s2) == ARKMPC_OK);
            CHECK(arkmpc_group_hostmul_finish(s2, 7, key[0], hde[1], hout[0]) == ARKMPC_ERR_BAD_ARG);       /* ends the session */
            CHECK(arkmpc_group_hostmul_begin(grp[0], n, hx[0], hy[0], ha[0], hb[0], hc[0], hde[0], &s2) == ARKMPC_OK);
            CHECK(arkmpc_group_hostmul_abort(s2) == ARKMPC_OK);
        }
        for (int p = 0; p < 2; ++p) {
            uint64_t* all[7] = {hx[p], hy[p], ha[p], hb[p], hc[p], hde[p], hout[p]};
            for (int k = 0; k < 7; ++k) { if (pinned) CHECK(arkmpc_host_free(all[k]) == ARKMPC_OK); else free(all[k]); }
        }
    }
    /* misuse is a status code */
    {
        shards_t bad;
        memset(bad, 0, sizeof bad);
        CHECK(arkmpc_group_beaver_mask(grp[0], 7, n, CS(bad), CS(bad), CS(bad), CS(bad), bad) == ARKMPC_ERR_BAD_ARG);
        if (n >= (size_t)G) CHECK(arkmpc_group_beaver_mask(grp[0], ARKMPC_LAYOUT_AOS, n, CS(bad), CS(bad), CS(bad), CS(bad), bad) == ARKMPC_ERR_BAD_ARG);
        CHECK(arkmpc_group_gather(grp[0], n, 2, 4, CS(bad), G, NULL) == ARKMPC_ERR_BAD_ARG);
        CHECK(arkmpc_group_shard_range(grp[0], n, G, NULL, NULL) == ARKMPC_ERR_BAD_ARG);
        int none[1] = {99};
        arkmpc_group* g2 = NULL;
        CHECK(arkmpc_group_create(ARKMPC_BN254_FR, 1, none, &g2) == ARKMPC_ERR_BAD_ARG && g2 == NULL);
        CHECK(arkmpc_group_create(ARKMPC_BN254_FR, 0, devs, &g2) == ARKMPC_ERR_BAD_ARG);
    }
    for (int p = 0; p < 2; ++p) { free(x[p]); free(y[p]); free(a[p]); free(b[p]); free(c[p]); free(key[p]); free(want_de[p]); free(want[p]); CHECK(arkmpc_group_destroy(grp[p]) == 0); grp[p] = NULL; }
    free(got);
    CHECK(arkmpc_ctx_destroy(ctx) == 0);
    printf("  beaver batch_mul over %zu gates on %d members: AoS + split, host / member hand-over, gather / gathered K1 / all-gather / scatter, group sessions (pageable + pinned): bit-equal\n", n, G);
    return 0;
}

static int open_authenticated(size_t n, int G, const int* devs) {
    arkmpc_ctx* ctx = NULL;
    arkmpc_group* grp[2] = {NULL, NULL};
    const int F = ARKMPC_BLS12_381_FR;
    CHECK(arkmpc_ctx_create(F, devs[0], &ctx) == ARKMPC_OK);
    for (int p = 0; p < 2; ++p) CHECK(arkmpc_group_create(F, G, devs, &grp[p]) == ARKMPC_OK);
    uint64_t *sh[2], *key[2], *blind[2], *want_open[2], *want_chk[2], want_comm[2][4];
    uint64_t *got = zeros(4 * n), *mine = zeros(4 * n);
    for (int p = 0; p < 2; ++p) { sh[p] = fill(2 * n); key[p] = fill(1); blind[p] = fill(1); want_open[p] = zeros(4 * n); want_chk[p] = zeros(4 * n); }
    /* expectation on ONE context (random shares: the MACs are not valid, verification is exercised below with crafted chk vectors) */
    {
        void *dsh[2], *dmine[2], *dop, *dchk;
        for (int p = 0; p < 2; ++p) {
            CHECK(arkmpc_malloc(ctx, n * 64 + 16, &dsh[p]) == 0 && arkmpc_malloc(ctx, n * 32 + 16, &dmine[p]) == 0);
            CHECK(arkmpc_memcpy_h2d(ctx, dsh[p], sh[p], n * 64) == 0);
            CHECK(arkmpc_share_extract(ctx, n, dsh[p], dmine[p]) == ARKMPC_OK);
        }
        CHECK(arkmpc_malloc(ctx, n * 32 + 16, &dop) == 0 && arkmpc_malloc(ctx, n * 32 + 16, &dchk) == 0);
        for (int p = 0; p < 2; ++p) {
            CHECK(arkmpc_open_and_mac_check(ctx, n, key[p], dsh[p], dmine[1 - p], dop, dchk) == ARKMPC_OK);
            CHECK(arkmpc_memcpy_d2h(ctx, want_open[p], dop, n * 32) == 0 && arkmpc_memcpy_d2h(ctx, want_chk[p], dchk, n * 32) == 0);
            CHECK(arkmpc_commit_sha3(ctx, n, dchk, blind[p], want_comm[p]) == ARKMPC_OK);
        }
        arkmpc_free(ctx, dop); arkmpc_free(ctx, dchk);
        for (int p = 0; p < 2; ++p) { arkmpc_free(ctx, dsh[p]); arkmpc_free(ctx, dmine[p]); }
    }
    CHECK(memcmp(want_open[0], want_open[1], n * 32) == 0);
    for (int layout = ARKMPC_LAYOUT_AOS; layout <= ARKMPC_LAYOUT_SPLIT; ++layout) {
        const size_t segs = layout == ARKMPC_LAYOUT_SPLIT ? 2 : 1, ew = layout == ARKMPC_LAYOUT_SPLIT ? 4 : 8;
        shards_t ssh[2], smine[2], sop[2], schk[2];
        for (int p = 0; p < 2; ++p) {
            CHECK(arkmpc_group_malloc(grp[p], n, segs, ew, ssh[p]) == 0 && arkmpc_group_malloc(grp[p], n, 1, 4, smine[p]) == 0 &&
                  arkmpc_group_malloc(grp[p], n, 1, 4, sop[p]) == 0 && arkmpc_group_malloc(grp[p], n, 1, 4, schk[p]) == 0);
            CHECK(arkmpc_group_shares_from_host(grp[p], layout, n, sh[p], ssh[p]) == ARKMPC_OK);
            CHECK(arkmpc_group_share_extract(grp[p], layout, n, CS(ssh[p]), smine[p]) == ARKMPC_OK);
            CHECK(arkmpc_group_sync(grp[p]) == ARKMPC_OK);
        }
        for (int p = 0; p < 2; ++p) {
            uint64_t comm[4];
            CHECK(arkmpc_group_open_and_mac_check(grp[p], layout, n, key[p], CS(ssh[p]), CS(smine[1 - p]), sop[p], schk[p]) == ARKMPC_OK);
            CHECK(arkmpc_group_gather_d2h(grp[p], n, 1, 4, CS(sop[p]), got) == ARKMPC_OK && memcmp(got, want_open[p], n * 32) == 0);
            CHECK(arkmpc_group_gather_d2h(grp[p], n, 1, 4, CS(schk[p]), got) == ARKMPC_OK && memcmp(got, want_chk[p], n * 32) == 0);
            CHECK(arkmpc_group_commit_sha3(grp[p], n, CS(schk[p]), blind[p], comm) == ARKMPC_OK);
            CHECK(memcmp(comm, want_comm[p], 32) == 0);
        }
        /* K5 + AND-reduce: mine + peer == 0 everywhere, then with the LAST element (the last member's range) off by one */
        {
            shards_t sneg;
            int ok = -1;
            CHECK(arkmpc_group_malloc(grp[0], n, 1, 4, sneg) == 0);
            for (int m = 0; m < G; ++m) {
                size_t lo, cnt;
                CHECK(arkmpc_group_shard_range(grp[0], n, m, &lo, &cnt) == 0);
                if (cnt) CHECK(arkmpc_scalar_neg(arkmpc_group_ctx(grp[0], m), cnt, schk[0][m], sneg[m]) == ARKMPC_OK);
            }
            CHECK(arkmpc_group_mac_verify(grp[0], n, CS(schk[0]), CS(sneg), &ok) == ARKMPC_OK && ok == 1);
            if (n) {
                CHECK(arkmpc_group_gather_d2h(grp[0], n, 1, 4, CS(sneg), got) == ARKMPC_OK);
                got[4 * (n - 1)] ^= 1;                                   /* still < p: the low bit of a residue below 2^255 */
                CHECK(arkmpc_group_scatter_h2d(grp[0], n, 1, 4, got, sneg) == ARKMPC_OK);
                CHECK(arkmpc_group_mac_verify(grp[0], n, CS(schk[0]), CS(sneg), &ok) == ARKMPC_OK && ok == 0);
                got[4 * (n - 1)] ^= 1;
                CHECK(arkmpc_group_scatter_h2d(grp[0], n, 1, 4, got, sneg) == ARKMPC_OK);
                CHECK(arkmpc_group_mac_verify(grp[0], n, CS(schk[0]), CS(sneg), &ok) == ARKMPC_OK && ok == 1);      /* the flag was cleared */
            }
            CHECK(arkmpc_group_free(grp[0], sneg) == 0);
        }
        for (int p = 0; p < 2; ++p)
            CHECK(arkmpc_group_free(grp[p], ssh[p]) == 0 && arkmpc_group_free(grp[p], smine[p]) == 0 && arkmpc_group_free(grp[p], sop[p]) == 0 && arkmpc_group_free(grp[p], schk[p]) == 0);
    }
    for (int p = 0; p < 2; ++p) { free(sh[p]); free(key[p]); free(blind[p]); free(want_open[p]); free(want_chk[p]); CHECK(arkmpc_group_destroy(grp[p]) == 0); grp[p] = NULL; }
    free(got); free(mine);
    CHECK(arkmpc_ctx_destroy(ctx) == 0);
    printf("  open_authenticated_batch over %zu shares of BLS12-381 Fr on %d members: opened, chk, commitment, verify: bit-equal\n", n, G);
    return 0;
}

static int msm(size_t n, int G, const int* devs) {
    arkmpc_ctx* ctx = NULL;
    arkmpc_group* grp[2] = {NULL, NULL};
    CHECK(arkmpc_ctx_create(ARKMPC_BN254_FR, devs[0], &ctx) == ARKMPC_OK);
    CHECK(arkmpc_group_create(ARKMPC_BN254_FR, G, devs, &grp[0]) == ARKMPC_OK);
    uint64_t *s = fill(n), *t = fill(n);
    void *ds, *dt, *dp, *dout, *dxy, *dinf;
    CHECK(arkmpc_malloc(ctx, n * 32 + 16, &ds) == 0 && arkmpc_malloc(ctx, n * 32 + 16, &dt) == 0 && arkmpc_malloc(ctx, n * 96 + 16, &dp) == 0 &&
          arkmpc_malloc(ctx, 2 * 96, &dout) == 0 && arkmpc_malloc(ctx, 2 * 64, &dxy) == 0 && arkmpc_malloc(ctx, 16, &dinf) == 0);
    CHECK(arkmpc_memcpy_h2d(ctx, ds, s, n * 32) == 0 && arkmpc_memcpy_h2d(ctx, dt, t, n * 32) == 0);
    CHECK(arkmpc_g1_generator_mul(ctx, n, (const uint64_t*)dt, (uint64_t*)dp) == ARKMPC_OK);          /* points P_i = t_i G */
    CHECK(arkmpc_g1_msm(ctx, n, (const uint64_t*)dp, (const uint64_t*)ds, (uint64_t*)dout) == ARKMPC_OK);
    uint64_t* hp = zeros(12 * n);
    CHECK(arkmpc_memcpy_d2h(ctx, hp, dp, n * 96) == 0);
    shards_t sp, ss;
    uint64_t gsum[12], xy[16];
    unsigned char inf[2];
    CHECK(arkmpc_group_malloc(grp[0], n, 1, 12, sp) == 0 && arkmpc_group_malloc(grp[0], n, 1, 4, ss) == 0);
    CHECK(arkmpc_group_scatter_h2d(grp[0], n, 1, 12, hp, sp) == 0 && arkmpc_group_scatter_h2d(grp[0], n, 1, 4, s, ss) == 0);
    CHECK(arkmpc_group_g1_msm(grp[0], n, CS(sp), CS(ss), gsum) == ARKMPC_OK);
    CHECK(arkmpc_memcpy_h2d(ctx, (char*)dout + 96, gsum, 96) == 0);
    CHECK(arkmpc_g1_to_affine(ctx, 2, (const uint64_t*)dout, (uint64_t*)dxy, (uint8_t*)dinf) == ARKMPC_OK);   /* representatives differ: compare affine */
    CHECK(arkmpc_memcpy_d2h(ctx, xy, dxy, 128) == 0 && arkmpc_memcpy_d2h(ctx, inf, dinf, 2) == 0);
    CHECK(inf[0] == inf[1] && memcmp(xy, xy + 8, 64) == 0);
    CHECK(arkmpc_group_free(grp[0], sp) == 0 && arkmpc_group_free(grp[0], ss) == 0);
    arkmpc_free(ctx, ds); arkmpc_free(ctx, dt); arkmpc_free(ctx, dp); arkmpc_free(ctx, dout); arkmpc_free(ctx, dxy); arkmpc_free(ctx, dinf);
    free(s); free(t); free(hp);
    CHECK(arkmpc_group_destroy(grp[0]) == 0); grp[0] = NULL;
    CHECK(arkmpc_ctx_destroy(ctx) == 0);
    printf("  bucket MSM over %zu points on %d members == one context (affine)\n", n, G);
    return 0;
}

/* the same on Curve25519 (16-word extended points; fill() keeps values below 2^252 < l) */
static int ed_msm(size_t n, int G, const int* devs) {
    arkmpc_ctx* ctx = NULL;
    arkmpc_group* grp[2] = {NULL, NULL};
    CHECK(arkmpc_ctx_create(ARKMPC_CURVE25519_FR, devs[0], &ctx) == ARKMPC_OK);
    CHECK(arkmpc_group_create(ARKMPC_CURVE25519_FR, G, devs, &grp[0]) == ARKMPC_OK);
    uint64_t *s = fill(n), *t = fill(n);
    void *ds, *dt, *dp, *dout, *dxy;
    CHECK(arkmpc_malloc(ctx, n * 32 + 16, &ds) == 0 && arkmpc_malloc(ctx, n * 32 + 16, &dt) == 0 && arkmpc_malloc(ctx, n * 128 + 16, &dp) == 0 &&
          arkmpc_malloc(ctx, 2 * 128, &dout) == 0 && arkmpc_malloc(ctx, 2 * 64, &dxy) == 0);
    CHECK(arkmpc_memcpy_h2d(ctx, ds, s, n * 32) == 0 && arkmpc_memcpy_h2d(ctx, dt, t, n * 32) == 0);
    CHECK(arkmpc_ed_generator_mul(ctx, n, (const uint64_t*)dt, (uint64_t*)dp) == ARKMPC_OK);          /* points P_i = t_i B */
    CHECK(arkmpc_ed_msm(ctx, n, (const uint64_t*)dp, (const uint64_t*)ds, (uint64_t*)dout) == ARKMPC_OK);
    uint64_t* hp = zeros(16 * n);
    CHECK(arkmpc_memcpy_d2h(ctx, hp, dp, n * 128) == 0);
    shards_t sp, ss;
    uint64_t gsum[16], xy[16];
    CHECK(arkmpc_group_malloc(grp[0], n, 1, 16, sp) == 0 && arkmpc_group_malloc(grp[0], n, 1, 4, ss) == 0);
    CHECK(arkmpc_group_scatter_h2d(grp[0], n, 1, 16, hp, sp) == 0 && arkmpc_group_scatter_h2d(grp[0], n, 1, 4, s, ss) == 0);
    CHECK(arkmpc_group_ed_msm(grp[0], n, CS(sp), CS(ss), gsum) == ARKMPC_OK);
    CHECK(arkmpc_group_g1_msm(grp[0], n, CS(sp), CS(ss), gsum) == ARKMPC_ERR_UNSUPPORTED);                /* a Curve25519 group has no G1 */
    CHECK(arkmpc_memcpy_h2d(ctx, (char*)dout + 128, gsum, 128) == 0);
    CHECK(arkmpc_ed_to_affine(ctx, 2, (const uint64_t*)dout, (uint64_t*)dxy) == ARKMPC_OK);                /* representatives differ: compare affine */
    CHECK(arkmpc_memcpy_d2h(ctx, xy, dxy, 128) == 0);
    CHECK(memcmp(xy, xy + 8, 64) == 0);
    CHECK(arkmpc_group_free(grp[0], sp) == 0 && arkmpc_group_free(grp[0], ss) == 0);
    arkmpc_free(ctx, ds); arkmpc_free(ctx, dt); arkmpc_free(ctx, dp); arkmpc_free(ctx, dout); arkmpc_free(ctx, dxy);
    free(s); free(t); free(hp);
    CHECK(arkmpc_group_destroy(grp[0]) == 0); grp[0] = NULL;
    CHECK(arkmpc_ctx_destroy(ctx) == 0);
    printf("  Curve25519 bucket MSM over %zu points on %d members == one context (affine)\n", n, G);
    return 0;
}

int main(int argc, char** argv) {
    size_t n = argc > 1 ? (size_t)strtoull(argv[1], NULL, 10) : 100003;
    int G = argc > 2 ? atoi(argv[2]) : 4;
    int devs[MAXG];
    if (G < 1 || G > MAXG) { printf("G out of range\n"); return 2; }
    for (int m = 0; m < G; ++m) devs[m] = argc > 3 + m ? atoi(argv[3 + m]) : 0;
    if (arkmpc_device_count() <= 0) { printf("no device: status %d\n", ARKMPC_ERR_NO_DEVICE); return 3; }
    if (beaver(n, G, devs)) return 1;
    if (open_authenticated(n, G, devs)) return 1;
    if (msm(n < 50000 ? n : 50000, G, devs)) return 1;
    if (ed_msm(n < 50000 ? n : 50000, G, devs)) return 1;
    printf("group ok: n = %zu, %d members\n", n, G);
    return 0;
}
