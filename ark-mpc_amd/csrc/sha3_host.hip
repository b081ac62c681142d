// sha3_host.hip -- host side of the hash commitment (H1): SHA3-256 sponge + the two scalar
// conversions at its ends.  Replaces online-phase/src/commitment.rs:30-43 / :71-86, which call the
// `sha3` crate and Scalar::{to_bytes_be, from_be_bytes_mod_order} (scalar.rs:109-127).
// The sponge is sequential by definition (one Keccak-f[1600] per 136-byte block), so it stays on
// the CPU; the GPU produces the big-endian byte stream (K6) and the D2H copy overlaps hashing.
#include "arkmpc_internal.hpp"
#include <chrono>
#include <cstdlib>

namespace {

inline uint64_t rol(uint64_t x, int s) { return (x << s) | (x >> (64 - s)); }

const uint64_t RNDC[24] = {
    0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL, 0x000000000000808bULL,
    0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL, 0x000000000000008aULL, 0x0000000000000088ULL,
    0x0000000080008009ULL, 0x000000008000000aULL, 0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL,
    0x8000000000008003ULL, 0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
    0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};

// One permutation; lanes held in locals so the compiler keeps them in registers.
void keccak_f1600(uint64_t* A) {
    uint64_t a00 = A[0], a01 = A[1], a02 = A[2], a03 = A[3], a04 = A[4], a05 = A[5], a06 = A[6], a07 = A[7], a08 = A[8],
             a09 = A[9], a10 = A[10], a11 = A[11], a12 = A[12], a13 = A[13], a14 = A[14], a15 = A[15], a16 = A[16],
             a17 = A[17], a18 = A[18], a19 = A[19], a20 = A[20], a21 = A[21], a22 = A[22], a23 = A[23], a24 = A[24];
    for (int r = 0; r < 24; ++r) {
        // theta
        uint64_t c0 = a00 ^ a05 ^ a10 ^ a15 ^ a20, c1 = a01 ^ a06 ^ a11 ^ a16 ^ a21, c2 = a02 ^ a07 ^ a12 ^ a17 ^ a22,
                 c3 = a03 ^ a08 ^ a13 ^ a18 ^ a23, c4 = a04 ^ a09 ^ a14 ^ a19 ^ a24;
        uint64_t d0 = c4 ^ rol(c1, 1), d1 = c0 ^ rol(c2, 1), d2 = c1 ^ rol(c3, 1), d3 = c2 ^ rol(c4, 1), d4 = c3 ^ rol(c0, 1);
        a00 ^= d0; a05 ^= d0; a10 ^= d0; a15 ^= d0; a20 ^= d0;
        a01 ^= d1; a06 ^= d1; a11 ^= d1; a16 ^= d1; a21 ^= d1;
        a02 ^= d2; a07 ^= d2; a12 ^= d2; a17 ^= d2; a22 ^= d2;
        a03 ^= d3; a08 ^= d3; a13 ^= d3; a18 ^= d3; a23 ^= d3;
        a04 ^= d4; a09 ^= d4; a14 ^= d4; a19 ^= d4; a24 ^= d4;
        // rho + pi  (b[y][2x+3y] = rot(a[x][y]))
        uint64_t b00 = a00, b10 = rol(a01, 1), b20 = rol(a02, 62), b05 = rol(a03, 28), b15 = rol(a04, 27);
        uint64_t b16 = rol(a05, 36), b01 = rol(a06, 44), b11 = rol(a07, 6), b21 = rol(a08, 55), b06 = rol(a09, 20);
        uint64_t b07 = rol(a10, 3), b17 = rol(a11, 10), b02 = rol(a12, 43), b12 = rol(a13, 25), b22 = rol(a14, 39);
        uint64_t b23 = rol(a15, 41), b08 = rol(a16, 45), b18 = rol(a17, 15), b03 = rol(a18, 21), b13 = rol(a19, 8);
        uint64_t b14 = rol(a20, 18), b24 = rol(a21, 2), b09 = rol(a22, 61), b19 = rol(a23, 56), b04 = rol(a24, 14);
        // chi
        a00 = b00 ^ (~b01 & b02); a01 = b01 ^ (~b02 & b03); a02 = b02 ^ (~b03 & b04); a03 = b03 ^ (~b04 & b00); a04 = b04 ^ (~b00 & b01);
        a05 = b05 ^ (~b06 & b07); a06 = b06 ^ (~b07 & b08); a07 = b07 ^ (~b08 & b09); a08 = b08 ^ (~b09 & b05); a09 = b09 ^ (~b05 & b06);
        a10 = b10 ^ (~b11 & b12); a11 = b11 ^ (~b12 & b13); a12 = b12 ^ (~b13 & b14); a13 = b13 ^ (~b14 & b10); a14 = b14 ^ (~b10 & b11);
        a15 = b15 ^ (~b16 & b17); a16 = b16 ^ (~b17 & b18); a17 = b17 ^ (~b18 & b19); a18 = b18 ^ (~b19 & b15); a19 = b19 ^ (~b15 & b16);
        a20 = b20 ^ (~b21 & b22); a21 = b21 ^ (~b22 & b23); a22 = b22 ^ (~b23 & b24); a23 = b23 ^ (~b24 & b20); a24 = b24 ^ (~b20 & b21);
        // iota
        a00 ^= RNDC[r];
    }
    A[0] = a00; A[1] = a01; A[2] = a02; A[3] = a03; A[4] = a04; A[5] = a05; A[6] = a06; A[7] = a07; A[8] = a08; A[9] = a09;
    A[10] = a10; A[11] = a11; A[12] = a12; A[13] = a13; A[14] = a14; A[15] = a15; A[16] = a16; A[17] = a17; A[18] = a18;
    A[19] = a19; A[20] = a20; A[21] = a21; A[22] = a22; A[23] = a23; A[24] = a24;
}

inline void absorb136(uint64_t* st, const unsigned char* blk) {
    for (int i = 0; i < 17; ++i) {
        uint64_t w;
        memcpy(&w, blk + 8 * i, 8);  // little-endian host
        st[i] ^= w;
    }
    keccak_f1600(st);
}

// Inner loops from keccak_avx512.cpp.  Which one is fastest depends on the host (measured: the AVX-512 form wins 1.7x on a
// 2.1 GHz Xeon, the 64-bit form compiled with BMI wins on EPYC 9575F), so the first multi-block update times each candidate
// on 64 blocks and keeps the winner.  ARKMPC_KECCAK=portable|scalar|bmi|avx512|lanes|rows forces one.
extern "C" __attribute__((visibility("hidden"))) int arkmpc_cpu_has_avx512(void);
extern "C" __attribute__((visibility("hidden"))) int arkmpc_cpu_has_bmi(void);
extern "C" __attribute__((visibility("hidden"))) void arkmpc_keccak_absorb136_avx512(uint64_t st[25], const unsigned char* data, size_t nblocks);
extern "C" __attribute__((visibility("hidden"))) void arkmpc_keccak_absorb136_lanes(uint64_t st[25], const unsigned char* data, size_t nblocks);
extern "C" __attribute__((visibility("hidden"))) void arkmpc_keccak_absorb136_rows(uint64_t st[25], const unsigned char* data, size_t nblocks);
extern "C" __attribute__((visibility("hidden"))) void arkmpc_keccak_absorb136_scalar(uint64_t st[25], const unsigned char* data, size_t nblocks);
extern "C" __attribute__((visibility("hidden"))) void arkmpc_keccak_absorb136_bmi(uint64_t st[25], const unsigned char* data, size_t nblocks);
typedef void (*absorb_fn)(uint64_t*, const unsigned char*, size_t);
void absorb136_portable(uint64_t* st, const unsigned char* data, size_t nblocks) {
    for (size_t i = 0; i < nblocks; ++i) absorb136(st, data + 136 * i);
}
struct Picked { const char* name; absorb_fn fn; };
Picked pick_absorb() {
    struct Cand { const char* name; absorb_fn fn; bool ok; };
    const Cand cands[] = {{"portable", absorb136_portable, true}, {"scalar", arkmpc_keccak_absorb136_scalar, sizeof(void*) == 8},
                          {"bmi", arkmpc_keccak_absorb136_bmi, arkmpc_cpu_has_bmi() != 0}, {"avx512", arkmpc_keccak_absorb136_avx512, arkmpc_cpu_has_avx512() != 0},
                          {"lanes", arkmpc_keccak_absorb136_lanes, arkmpc_cpu_has_avx512() != 0},
                          {"rows", arkmpc_keccak_absorb136_rows, arkmpc_cpu_has_avx512() != 0}};
    if (const char* force = getenv("ARKMPC_KECCAK"))
        for (const Cand& c : cands) if (c.ok && !strcmp(force, c.name)) return Picked{c.name, c.fn};
    static unsigned char probe[512 * 136];                // (64 blocks x 3 runs picked a slower loop now and then on a busy host: config 5 end to end 1.22 vs 1.40 s)
    for (size_t i = 0; i < sizeof(probe); ++i) probe[i] = (unsigned char)(i * 131u + 7u);
    Picked best{"portable", absorb136_portable};
    double best_t = 1e300;
    for (const Cand& c : cands) {
        if (!c.ok) continue;
        uint64_t st[25] = {0};
        double t = 1e300;
        for (int rep = 0; rep < 7; ++rep) {
            const auto t0 = std::chrono::steady_clock::now();
            c.fn(st, probe, 512);
            const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (dt < t) t = dt;
        }
        if (t < best_t) { best_t = t; best = Picked{c.name, c.fn}; }
    }
    return best;
}
const Picked& picked_absorb() {
    static const Picked p = pick_absorb();
    return p;
}
absorb_fn absorb_blocks() { return picked_absorb().fn; }

template <int F> void to_be_t(const uint64_t m[4], unsigned char out[32]) {
    Fe c = fe_to_canonical<F>(fe_from_host(m));
    for (int i = 0; i < 8; ++i) {
        u32 w = c.v[7 - i];
        out[4 * i] = (unsigned char)(w >> 24); out[4 * i + 1] = (unsigned char)(w >> 16);
        out[4 * i + 2] = (unsigned char)(w >> 8); out[4 * i + 3] = (unsigned char)w;
    }
}
template <int F> void from_be_t(const unsigned char be[32], uint64_t out[4]) {
    Fe v;
    for (int i = 0; i < 8; ++i)
        v.v[7 - i] = ((u32)be[4 * i] << 24) | ((u32)be[4 * i + 1] << 16) | ((u32)be[4 * i + 2] << 8) | (u32)be[4 * i + 3];
    Fe m = fe_from_canonical<F>(fe_reduce_once_loop<F>(v));
    for (int i = 0; i < 4; ++i) out[i] = (uint64_t)m.v[2 * i] | ((uint64_t)m.v[2 * i + 1] << 32);
}

}  // namespace

extern "C" const char* arkmpc_sha3_loop(void) { return picked_absorb().name; }

void sha3_256_init(Sha3State* s) { memset(s, 0, sizeof(*s)); }

void sha3_256_update(Sha3State* s, const unsigned char* msg, size_t len) {
    if (s->fill) {
        size_t take = 136 - s->fill;
        if (take > len) take = len;
        memcpy(s->buf + s->fill, msg, take);
        s->fill += take; msg += take; len -= take;
        if (s->fill == 136) { absorb136(s->st, s->buf); s->fill = 0; }
    }
    if (len >= 136) {
        const size_t nb = len / 136;
        absorb_blocks()(s->st, msg, nb);
        msg += nb * 136; len -= nb * 136;
    }
    if (len) { memcpy(s->buf, msg, len); s->fill = len; }
}

void sha3_256_final(Sha3State* s, unsigned char out[32]) {
    memset(s->buf + s->fill, 0, 136 - s->fill);
    s->buf[s->fill] ^= 0x06;   // SHA3 domain separation + first pad bit
    s->buf[135] ^= 0x80;       // last pad bit
    absorb136(s->st, s->buf);
    memcpy(out, s->st, 32);
}

void host_to_bytes_be(int field_id, const uint64_t mont[4], unsigned char out[32]) {
    switch (field_id) {
        case 0: to_be_t<0>(mont, out); break;
        case 1: to_be_t<1>(mont, out); break;
        case 2: to_be_t<2>(mont, out); break;
        case 3: to_be_t<3>(mont, out); break;
        default: to_be_t<4>(mont, out); break;
    }
}
void host_from_be_bytes_mod_order(int field_id, const unsigned char be[32], uint64_t out_mont[4]) {
    switch (field_id) {
        case 0: from_be_t<0>(be, out_mont); break;
        case 1: from_be_t<1>(be, out_mont); break;
        case 2: from_be_t<2>(be, out_mont); break;
        case 3: from_be_t<3>(be, out_mont); break;
        default: from_be_t<4>(be, out_mont); break;
    }
}
