/* bench_port.c -- CPU restatement of the reference's two latency-bound criterion benches, on the oracle's arithmetic (TEST / BENCH
 * INFRASTRUCTURE: a reported baseline beside the GPU mirror's numbers, never part of the product).
 *   batch_ops       online-phase/benches/batch_ops.rs:20-39: share x, share y (fabric.rs:578-600), batch_mul as the literal 9 passes
 *                   (authenticated_scalar.rs:848-879), open_authenticated_batch (:278-354) incl. both SHA3 commitments per party
 *   mul_throughput  benches/circuit_mul_throughput.rs:24-36: n SEQUENTIAL squarings of one shared value, then an open
 * with the reference's PartyIDBeaverSource (offline_prep.rs:88-170).  ONE thread executes both parties' arithmetic step by step, the
 * "network" is a memcpy; the reference runs the parties concurrently, so its per-bench time corresponds to seconds_per_party here
 * (= the total / 2) plus its executor and channel overheads, which this port does not have: it is a LOWER bound for the reference.
 * usage: bench_port <batch_ops|mul_throughput> <n> [iters]   -> one JSON line */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "ark_oracle.h"

static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
static u64* buf(size_t words) { u64* p = (u64*)calloc(words ? words : 1, 8); if (!p) { fprintf(stderr, "oom\n"); exit(1); } return p; }
static void from_u64(const ora_field* f, u64 v, u64 out[4]) { u64 c[4] = {v, 0, 0, 0}; ora_fp_from_canonical(f, c, out); }
static void rep(u64* dst, size_t n, const u64* rec, size_t words) { for (size_t i = 0; i < n; ++i) memcpy(dst + words * i, rec, words * 8); }

int main(int argc, char** argv) {
    if (argc < 3) { fprintf(stderr, "usage: %s <batch_ops|mul_throughput> <n> [iters]\n", argv[0]); return 2; }
    const char* bench = argv[1];
    const size_t n = strtoull(argv[2], NULL, 10);
    const int iters = argc > 3 ? atoi(argv[3]) : 5;
    const int fid = 0;
    const ora_field* f = ora_get_field(fid);
    u64 c[7][4];
    for (u64 v = 0; v <= 6; ++v) from_u64(f, v, c[v]);
    /* PartyIDBeaverSource: key share = party id; a = 2, b = 3, c = 6 split as (1,1) (3,0) (2,4); MACs = key_share * value */
    u64 key[2][4], ta[2][8], tb[2][8], tc[2][8], mask_local[2][8], mask_cp[2][8];
    for (int p = 0; p < 2; ++p) {
        memcpy(key[p], c[p], 32);
        memcpy(ta[p], c[1], 32); memcpy(ta[p] + 4, c[p * 2], 32);
        memcpy(tb[p], p ? c[0] : c[3], 32); memcpy(tb[p] + 4, c[p * 3], 32);
        memcpy(tc[p], p ? c[4] : c[2], 32); memcpy(tc[p] + 4, c[p * 6], 32);
        memcpy(mask_local[p], c[3 * p], 32); memcpy(mask_local[p] + 4, c[3 * p], 32);     /* offline_prep.rs:112-127 */
        memcpy(mask_cp[p], c[3 * p], 32); memcpy(mask_cp[p] + 4, c[3 * p], 32);
    }
    double best = 1e300;
    if (!strcmp(bench, "batch_ops")) {
        u64 *x = buf(4 * n), *y = buf(4 * n), *masks = buf(4 * n), *masked = buf(4 * n);
        for (size_t i = 0; i < 4 * n; ++i) { x[i] = 0x9E3779B97F4A7C15ull * (i + 1); y[i] = ~x[i] * 7; if ((i & 3) == 3) { x[i] >>= 8; y[i] >>= 9; } }
        u64 *sx[2], *sy[2], *A[2], *B[2], *C[2], *de[2], *res[2], *scr = buf(64 * n), *mine[2], *opened[2], *chk[2], *tmp = buf(8 * n);
        for (int p = 0; p < 2; ++p) { sx[p] = buf(8 * n); sy[p] = buf(8 * n); A[p] = buf(8 * n); B[p] = buf(8 * n); C[p] = buf(8 * n); de[p] = buf(8 * n); res[p] = buf(8 * n);
                                      mine[p] = buf(4 * n); opened[p] = buf(4 * n); chk[p] = buf(4 * n); }
        for (int it = 0; it < iters + 1; ++it) {
            const double t0 = now();
            /* share x and y from party 0: masked = v - mask (sender), both: mask_share.add_public(masked) */
            u64* vals[2] = {x, y}; u64** dst[2] = {sx, sy};
            for (int k = 0; k < 2; ++k) {
                rep(masks, n, c[3], 4);
                ora_scalar_batch_sub(fid, n, vals[k], masks, masked);
                for (int p = 0; p < 2; ++p) { rep(tmp, n, p == 0 ? mask_local[0] : mask_cp[1], 8); ora_share_batch_add_public(fid, n, p, key[p], tmp, masked, dst[k][p]); }
            }
            for (int p = 0; p < 2; ++p) { rep(A[p], n, ta[p], 8); rep(B[p], n, tb[p], 8); rep(C[p], n, tc[p], 8); }
            for (int p = 0; p < 2; ++p) ora_beaver_mask(fid, n, sx[p], sy[p], A[p], B[p], de[p]);              /* what each party sends */
            for (int p = 0; p < 2; ++p) ora_batch_mul_9pass_local(fid, n, p, key[p], sx[p], sy[p], A[p], B[p], C[p], de[1 - p], tmp, res[p], scr);
            /* open_authenticated_batch */
            for (int p = 0; p < 2; ++p) for (size_t i = 0; i < n; ++i) memcpy(mine[p] + 4 * i, res[p] + 8 * i, 32);
            u64 comm[2][4], re[2][4], bl[2][4];
            for (int p = 0; p < 2; ++p) {
                from_u64(f, 77 + p, bl[p]);
                ora_open_combine(fid, n, mine[p], mine[1 - p], opened[p]);
                ora_mac_check_shares(fid, n, key[p], opened[p], res[p], chk[p]);
                ora_commit_scalars(fid, n, chk[p], bl[p], comm[p]);
            }
            int ok = 1;
            for (int p = 0; p < 2; ++p) {
                ora_commit_scalars(fid, n, chk[1 - p], bl[1 - p], re[p]);
                ok &= memcmp(re[p], comm[1 - p], 32) == 0 && ora_mac_verify(fid, n, chk[p], chk[1 - p]);
            }
            const double s = now() - t0;
            if (!ok) { fprintf(stderr, "MAC check failed\n"); return 1; }
            if (it && s < best) best = s;
        }
    } else if (!strcmp(bench, "mul_throughput")) {
        for (int it = 0; it < iters + 1; ++it) {
            u64 r[2][8], d[2][8], out[2][8], scr[64], tmp[8], one_masked[4], msk[4];
            memcpy(msk, c[3], 32);
            ora_scalar_batch_sub(fid, 1, c[1], msk, one_masked);                                   /* share_scalar(1, PARTY0) */
            for (int p = 0; p < 2; ++p) ora_share_batch_add_public(fid, 1, p, key[p], p == 0 ? mask_local[0] : mask_cp[1], one_masked, r[p]);
            const double t0 = now();
            for (size_t k = 0; k < n; ++k) {
                for (int p = 0; p < 2; ++p) ora_beaver_mask(fid, 1, r[p], r[p], ta[p], tb[p], d[p]);
                for (int p = 0; p < 2; ++p) ora_batch_mul_9pass_local(fid, 1, p, key[p], r[p], r[p], ta[p], tb[p], tc[p], d[1 - p], tmp, out[p], scr);
                memcpy(r, out, sizeof r);
            }
            u64 o[4];
            ora_open_combine(fid, 1, r[0], r[1], o);
            const double s = now() - t0;
            u64 cn[4]; ora_fp_to_canonical(f, o, cn);
            if (cn[0] != 1 || cn[1] | cn[2] | cn[3]) { fprintf(stderr, "1^(2^n) != 1\n"); return 1; }
            if (it && s < best) best = s;
        }
    } else { fprintf(stderr, "unknown bench\n"); return 2; }
    printf("{\"bench\": \"%s\", \"n\": %zu, \"iters\": %d, \"impl\": \"cpu port (oracle/bench_port.c, one thread runs both parties)\", \"seconds_both_parties\": %.6g, "
           "\"seconds_per_party\": %.6g, \"elements_per_s\": %.6g}\n", bench, n, iters, best, best / 2, (double)n / (best / 2));
    return 0;
}
