// keccak_lanes_probe.cpp -- host probe: Keccak-f[1600] with ONE LANE PER VECTOR REGISTER (AVX-512VL on xmm: vpternlogq for the
// five-input column parity, theta's apply and chi; vprolq for every rotation; 32 registers, so no spills of the 25 lanes) against the
// 64-bit BMI form the library picks on EPYC 9575F.  Build + run:  g++ -O3 -std=c++17 -o /tmp/kl probes/keccak_lanes_probe.cpp && /tmp/kl
// The H1 sponge of config 5 is sequential (commitment.rs:36-40): the rate of ONE core is the floor of open_authenticated_batch end to end.
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <immintrin.h>
#include <vector>

static const uint64_t RC[24] = {
    0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL, 0x000000000000808bULL,
    0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL, 0x000000000000008aULL, 0x0000000000000088ULL,
    0x0000000080008009ULL, 0x000000008000000aULL, 0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL,
    0x8000000000008003ULL, 0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
    0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};

#define T512 __attribute__((target("avx512f,avx512vl")))
typedef __m128i V;
#define X3(a, b, c) _mm_ternarylogic_epi64(a, b, c, 0x96)
#define CHI(a, b, c) _mm_ternarylogic_epi64(a, b, c, 0xD2)
#define ROL(a, n) _mm_rol_epi64(a, n)
#define LD(p) _mm_loadl_epi64((const __m128i*)(p))

#define ROUND(rc)                                                                                                              \
    {                                                                                                                          \
        V c0 = X3(X3(a00, a05, a10), a15, a20), c1 = X3(X3(a01, a06, a11), a16, a21), c2 = X3(X3(a02, a07, a12), a17, a22),     \
          c3 = X3(X3(a03, a08, a13), a18, a23), c4 = X3(X3(a04, a09, a14), a19, a24);                                           \
        V r0 = ROL(c0, 1), r1 = ROL(c1, 1), r2 = ROL(c2, 1), r3 = ROL(c3, 1), r4 = ROL(c4, 1);                                  \
        V b00 = X3(a00, c4, r1), b10 = ROL(X3(a01, c0, r2), 1), b20 = ROL(X3(a02, c1, r3), 62), b05 = ROL(X3(a03, c2, r4), 28), \
          b15 = ROL(X3(a04, c3, r0), 27);                                                                                      \
        V b16 = ROL(X3(a05, c4, r1), 36), b01 = ROL(X3(a06, c0, r2), 44), b11 = ROL(X3(a07, c1, r3), 6),                        \
          b21 = ROL(X3(a08, c2, r4), 55), b06 = ROL(X3(a09, c3, r0), 20);                                                      \
        V b07 = ROL(X3(a10, c4, r1), 3), b17 = ROL(X3(a11, c0, r2), 10), b02 = ROL(X3(a12, c1, r3), 43),                        \
          b12 = ROL(X3(a13, c2, r4), 25), b22 = ROL(X3(a14, c3, r0), 39);                                                      \
        V b23 = ROL(X3(a15, c4, r1), 41), b08 = ROL(X3(a16, c0, r2), 45), b18 = ROL(X3(a17, c1, r3), 15),                       \
          b03 = ROL(X3(a18, c2, r4), 21), b13 = ROL(X3(a19, c3, r0), 8);                                                       \
        V b14 = ROL(X3(a20, c4, r1), 18), b24 = ROL(X3(a21, c0, r2), 2), b09 = ROL(X3(a22, c1, r3), 61),                        \
          b19 = ROL(X3(a23, c2, r4), 56), b04 = ROL(X3(a24, c3, r0), 14);                                                      \
        a00 = _mm_xor_si128(CHI(b00, b01, b02), LD(&(rc))); a01 = CHI(b01, b02, b03); a02 = CHI(b02, b03, b04);                 \
        a03 = CHI(b03, b04, b00); a04 = CHI(b04, b00, b01);                                                                    \
        a05 = CHI(b05, b06, b07); a06 = CHI(b06, b07, b08); a07 = CHI(b07, b08, b09); a08 = CHI(b08, b09, b05); a09 = CHI(b09, b05, b06); \
        a10 = CHI(b10, b11, b12); a11 = CHI(b11, b12, b13); a12 = CHI(b12, b13, b14); a13 = CHI(b13, b14, b10); a14 = CHI(b14, b10, b11); \
        a15 = CHI(b15, b16, b17); a16 = CHI(b16, b17, b18); a17 = CHI(b17, b18, b19); a18 = CHI(b18, b19, b15); a19 = CHI(b19, b15, b16); \
        a20 = CHI(b20, b21, b22); a21 = CHI(b21, b22, b23); a22 = CHI(b22, b23, b24); a23 = CHI(b23, b24, b20); a24 = CHI(b24, b20, b21); \
    }

T512 void absorb136_lanes(uint64_t st[25], const unsigned char* data, size_t nblocks) {
    V a00 = LD(st + 0), a01 = LD(st + 1), a02 = LD(st + 2), a03 = LD(st + 3), a04 = LD(st + 4), a05 = LD(st + 5), a06 = LD(st + 6),
      a07 = LD(st + 7), a08 = LD(st + 8), a09 = LD(st + 9), a10 = LD(st + 10), a11 = LD(st + 11), a12 = LD(st + 12), a13 = LD(st + 13),
      a14 = LD(st + 14), a15 = LD(st + 15), a16 = LD(st + 16), a17 = LD(st + 17), a18 = LD(st + 18), a19 = LD(st + 19), a20 = LD(st + 20),
      a21 = LD(st + 21), a22 = LD(st + 22), a23 = LD(st + 23), a24 = LD(st + 24);
    for (size_t blk = 0; blk < nblocks; ++blk, data += 136) {
#define AB(v, i) v = _mm_xor_si128(v, LD(data + 8 * (i)))
        AB(a00, 0); AB(a01, 1); AB(a02, 2); AB(a03, 3); AB(a04, 4); AB(a05, 5); AB(a06, 6); AB(a07, 7); AB(a08, 8); AB(a09, 9);
        AB(a10, 10); AB(a11, 11); AB(a12, 12); AB(a13, 13); AB(a14, 14); AB(a15, 15); AB(a16, 16);
#undef AB
        for (int r = 0; r < 24; r += 2) {
            ROUND(RC[r]);
            ROUND(RC[r + 1]);
        }
    }
#define ST(v, i) _mm_storel_epi64((__m128i*)(st + (i)), v)
    ST(a00, 0); ST(a01, 1); ST(a02, 2); ST(a03, 3); ST(a04, 4); ST(a05, 5); ST(a06, 6); ST(a07, 7); ST(a08, 8); ST(a09, 9);
    ST(a10, 10); ST(a11, 11); ST(a12, 12); ST(a13, 13); ST(a14, 14); ST(a15, 15); ST(a16, 16); ST(a17, 17); ST(a18, 18); ST(a19, 19);
    ST(a20, 20); ST(a21, 21); ST(a22, 22); ST(a23, 23); ST(a24, 24);
#undef ST
}

// row-wise schedules of the same round (five B values live at a time instead of twenty-five): ROUND2 with theta's D in registers, ROUND3 with the three-input fold
#define ROUND2(rc) { \
  V c0 = X3(X3(a00, a05, a10), a15, a20), c1 = X3(X3(a01, a06, a11), a16, a21), c2 = X3(X3(a02, a07, a12), a17, a22), c3 = X3(X3(a03, a08, a13), a18, a23), c4 = X3(X3(a04, a09, a14), a19, a24); \
  V d0 = _mm_xor_si128(c4, ROL(c1, 1)), d1 = _mm_xor_si128(c0, ROL(c2, 1)), d2 = _mm_xor_si128(c1, ROL(c3, 1)), d3 = _mm_xor_si128(c2, ROL(c4, 1)), d4 = _mm_xor_si128(c3, ROL(c0, 1)); \
  { V b0 = _mm_xor_si128(a00, d0), b1 = ROL(_mm_xor_si128(a06, d1), 44), b2 = ROL(_mm_xor_si128(a12, d2), 43), b3 = ROL(_mm_xor_si128(a18, d3), 21), b4 = ROL(_mm_xor_si128(a24, d4), 14); \
    n00 = CHI(b0, b1, b2); n01 = CHI(b1, b2, b3); n02 = CHI(b2, b3, b4); n03 = CHI(b3, b4, b0); n04 = CHI(b4, b0, b1); } \
  { V b0 = ROL(_mm_xor_si128(a03, d3), 28), b1 = ROL(_mm_xor_si128(a09, d4), 20), b2 = ROL(_mm_xor_si128(a10, d0), 3), b3 = ROL(_mm_xor_si128(a16, d1), 45), b4 = ROL(_mm_xor_si128(a22, d2), 61); \
    n05 = CHI(b0, b1, b2); n06 = CHI(b1, b2, b3); n07 = CHI(b2, b3, b4); n08 = CHI(b3, b4, b0); n09 = CHI(b4, b0, b1); } \
  { V b0 = ROL(_mm_xor_si128(a01, d1), 1), b1 = ROL(_mm_xor_si128(a07, d2), 6), b2 = ROL(_mm_xor_si128(a13, d3), 25), b3 = ROL(_mm_xor_si128(a19, d4), 8), b4 = ROL(_mm_xor_si128(a20, d0), 18); \
    n10 = CHI(b0, b1, b2); n11 = CHI(b1, b2, b3); n12 = CHI(b2, b3, b4); n13 = CHI(b3, b4, b0); n14 = CHI(b4, b0, b1); } \
  { V b0 = ROL(_mm_xor_si128(a04, d4), 27), b1 = ROL(_mm_xor_si128(a05, d0), 36), b2 = ROL(_mm_xor_si128(a11, d1), 10), b3 = ROL(_mm_xor_si128(a17, d2), 15), b4 = ROL(_mm_xor_si128(a23, d3), 56); \
    n15 = CHI(b0, b1, b2); n16 = CHI(b1, b2, b3); n17 = CHI(b2, b3, b4); n18 = CHI(b3, b4, b0); n19 = CHI(b4, b0, b1); } \
  { V b0 = ROL(_mm_xor_si128(a02, d2), 62), b1 = ROL(_mm_xor_si128(a08, d3), 55), b2 = ROL(_mm_xor_si128(a14, d4), 39), b3 = ROL(_mm_xor_si128(a15, d0), 41), b4 = ROL(_mm_xor_si128(a21, d1), 2); \
    n20 = CHI(b0, b1, b2); n21 = CHI(b1, b2, b3); n22 = CHI(b2, b3, b4); n23 = CHI(b3, b4, b0); n24 = CHI(b4, b0, b1); } \
  n00 = _mm_xor_si128(n00, LD(&(rc))); \
  a00 = n00; a01 = n01; a02 = n02; a03 = n03; a04 = n04; a05 = n05; a06 = n06; a07 = n07; a08 = n08; a09 = n09; a10 = n10; a11 = n11; a12 = n12; a13 = n13; a14 = n14; a15 = n15; a16 = n16; a17 = n17; a18 = n18; a19 = n19; a20 = n20; a21 = n21; a22 = n22; a23 = n23; a24 = n24; \
}
#define ROUND3(rc) { \
  V c0 = X3(X3(a00, a05, a10), a15, a20), c1 = X3(X3(a01, a06, a11), a16, a21), c2 = X3(X3(a02, a07, a12), a17, a22), c3 = X3(X3(a03, a08, a13), a18, a23), c4 = X3(X3(a04, a09, a14), a19, a24); \
  V r0 = ROL(c0, 1), r1 = ROL(c1, 1), r2 = ROL(c2, 1), r3 = ROL(c3, 1), r4 = ROL(c4, 1); \
  { V b0 = X3(a00, c4, r1), b1 = ROL(X3(a06, c0, r2), 44), b2 = ROL(X3(a12, c1, r3), 43), b3 = ROL(X3(a18, c2, r4), 21), b4 = ROL(X3(a24, c3, r0), 14); \
    n00 = CHI(b0, b1, b2); n01 = CHI(b1, b2, b3); n02 = CHI(b2, b3, b4); n03 = CHI(b3, b4, b0); n04 = CHI(b4, b0, b1); } \
  { V b0 = ROL(X3(a03, c2, r4), 28), b1 = ROL(X3(a09, c3, r0), 20), b2 = ROL(X3(a10, c4, r1), 3), b3 = ROL(X3(a16, c0, r2), 45), b4 = ROL(X3(a22, c1, r3), 61); \
    n05 = CHI(b0, b1, b2); n06 = CHI(b1, b2, b3); n07 = CHI(b2, b3, b4); n08 = CHI(b3, b4, b0); n09 = CHI(b4, b0, b1); } \
  { V b0 = ROL(X3(a01, c0, r2), 1), b1 = ROL(X3(a07, c1, r3), 6), b2 = ROL(X3(a13, c2, r4), 25), b3 = ROL(X3(a19, c3, r0), 8), b4 = ROL(X3(a20, c4, r1), 18); \
    n10 = CHI(b0, b1, b2); n11 = CHI(b1, b2, b3); n12 = CHI(b2, b3, b4); n13 = CHI(b3, b4, b0); n14 = CHI(b4, b0, b1); } \
  { V b0 = ROL(X3(a04, c3, r0), 27), b1 = ROL(X3(a05, c4, r1), 36), b2 = ROL(X3(a11, c0, r2), 10), b3 = ROL(X3(a17, c1, r3), 15), b4 = ROL(X3(a23, c2, r4), 56); \
    n15 = CHI(b0, b1, b2); n16 = CHI(b1, b2, b3); n17 = CHI(b2, b3, b4); n18 = CHI(b3, b4, b0); n19 = CHI(b4, b0, b1); } \
  { V b0 = ROL(X3(a02, c1, r3), 62), b1 = ROL(X3(a08, c2, r4), 55), b2 = ROL(X3(a14, c3, r0), 39), b3 = ROL(X3(a15, c4, r1), 41), b4 = ROL(X3(a21, c0, r2), 2); \
    n20 = CHI(b0, b1, b2); n21 = CHI(b1, b2, b3); n22 = CHI(b2, b3, b4); n23 = CHI(b3, b4, b0); n24 = CHI(b4, b0, b1); } \
  n00 = _mm_xor_si128(n00, LD(&(rc))); \
  a00 = n00; a01 = n01; a02 = n02; a03 = n03; a04 = n04; a05 = n05; a06 = n06; a07 = n07; a08 = n08; a09 = n09; a10 = n10; a11 = n11; a12 = n12; a13 = n13; a14 = n14; a15 = n15; a16 = n16; a17 = n17; a18 = n18; a19 = n19; a20 = n20; a21 = n21; a22 = n22; a23 = n23; a24 = n24; \
}
T512 void absorb136_rows_d(uint64_t st[25], const unsigned char* data, size_t nblocks) {
    V a00 = LD(st + 0), a01 = LD(st + 1), a02 = LD(st + 2), a03 = LD(st + 3), a04 = LD(st + 4), a05 = LD(st + 5), a06 = LD(st + 6),
      a07 = LD(st + 7), a08 = LD(st + 8), a09 = LD(st + 9), a10 = LD(st + 10), a11 = LD(st + 11), a12 = LD(st + 12), a13 = LD(st + 13),
      a14 = LD(st + 14), a15 = LD(st + 15), a16 = LD(st + 16), a17 = LD(st + 17), a18 = LD(st + 18), a19 = LD(st + 19), a20 = LD(st + 20),
      a21 = LD(st + 21), a22 = LD(st + 22), a23 = LD(st + 23), a24 = LD(st + 24);
    V n00, n01, n02, n03, n04, n05, n06, n07, n08, n09, n10, n11, n12, n13, n14, n15, n16, n17, n18, n19, n20, n21, n22, n23, n24;
    for (size_t blk = 0; blk < nblocks; ++blk, data += 136) {
#define AB(v, i) v = _mm_xor_si128(v, LD(data + 8 * (i)))
        AB(a00, 0); AB(a01, 1); AB(a02, 2); AB(a03, 3); AB(a04, 4); AB(a05, 5); AB(a06, 6); AB(a07, 7); AB(a08, 8); AB(a09, 9);
        AB(a10, 10); AB(a11, 11); AB(a12, 12); AB(a13, 13); AB(a14, 14); AB(a15, 15); AB(a16, 16);
#undef AB
        for (int r = 0; r < 24; r += 2) {
            ROUND2(RC[r]);
            ROUND2(RC[r + 1]);
        }
    }
#define ST(v, i) _mm_storel_epi64((__m128i*)(st + (i)), v)
    ST(a00, 0); ST(a01, 1); ST(a02, 2); ST(a03, 3); ST(a04, 4); ST(a05, 5); ST(a06, 6); ST(a07, 7); ST(a08, 8); ST(a09, 9);
    ST(a10, 10); ST(a11, 11); ST(a12, 12); ST(a13, 13); ST(a14, 14); ST(a15, 15); ST(a16, 16); ST(a17, 17); ST(a18, 18); ST(a19, 19);
    ST(a20, 20); ST(a21, 21); ST(a22, 22); ST(a23, 23); ST(a24, 24);
#undef ST
}

T512 void absorb136_rows_fold(uint64_t st[25], const unsigned char* data, size_t nblocks) {
    V a00 = LD(st + 0), a01 = LD(st + 1), a02 = LD(st + 2), a03 = LD(st + 3), a04 = LD(st + 4), a05 = LD(st + 5), a06 = LD(st + 6),
      a07 = LD(st + 7), a08 = LD(st + 8), a09 = LD(st + 9), a10 = LD(st + 10), a11 = LD(st + 11), a12 = LD(st + 12), a13 = LD(st + 13),
      a14 = LD(st + 14), a15 = LD(st + 15), a16 = LD(st + 16), a17 = LD(st + 17), a18 = LD(st + 18), a19 = LD(st + 19), a20 = LD(st + 20),
      a21 = LD(st + 21), a22 = LD(st + 22), a23 = LD(st + 23), a24 = LD(st + 24);
    V n00, n01, n02, n03, n04, n05, n06, n07, n08, n09, n10, n11, n12, n13, n14, n15, n16, n17, n18, n19, n20, n21, n22, n23, n24;
    for (size_t blk = 0; blk < nblocks; ++blk, data += 136) {
#define AB(v, i) v = _mm_xor_si128(v, LD(data + 8 * (i)))
        AB(a00, 0); AB(a01, 1); AB(a02, 2); AB(a03, 3); AB(a04, 4); AB(a05, 5); AB(a06, 6); AB(a07, 7); AB(a08, 8); AB(a09, 9);
        AB(a10, 10); AB(a11, 11); AB(a12, 12); AB(a13, 13); AB(a14, 14); AB(a15, 15); AB(a16, 16);
#undef AB
        for (int r = 0; r < 24; r += 2) {
            ROUND3(RC[r]);
            ROUND3(RC[r + 1]);
        }
    }
#define ST(v, i) _mm_storel_epi64((__m128i*)(st + (i)), v)
    ST(a00, 0); ST(a01, 1); ST(a02, 2); ST(a03, 3); ST(a04, 4); ST(a05, 5); ST(a06, 6); ST(a07, 7); ST(a08, 8); ST(a09, 9);
    ST(a10, 10); ST(a11, 11); ST(a12, 12); ST(a13, 13); ST(a14, 14); ST(a15, 15); ST(a16, 16); ST(a17, 17); ST(a18, 18); ST(a19, 19);
    ST(a20, 20); ST(a21, 21); ST(a22, 22); ST(a23, 23); ST(a24, 24);
#undef ST
}

static inline uint64_t rol64(uint64_t x, int s) { return (x << s) | (x >> (64 - s)); }
__attribute__((target("bmi,bmi2"))) void absorb136_bmi(uint64_t* A, const unsigned char* data, size_t nblocks) {
    uint64_t a[25];
    memcpy(a, A, 200);
    for (size_t blk = 0; blk < nblocks; ++blk, data += 136) {
        uint64_t w[17];
        memcpy(w, data, 136);
        for (int i = 0; i < 17; ++i) a[i] ^= w[i];
        for (int r = 0; r < 24; ++r) {
            uint64_t c[5], d[5], b[25];
            for (int x = 0; x < 5; ++x) c[x] = a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20];
            for (int x = 0; x < 5; ++x) d[x] = c[(x + 4) % 5] ^ rol64(c[(x + 1) % 5], 1);
            static const int rot[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};
            for (int y = 0; y < 5; ++y)
                for (int x = 0; x < 5; ++x) {
                    uint64_t v = a[x + 5 * y] ^ d[x];
                    int s = rot[x + 5 * y];
                    b[y + 5 * ((2 * x + 3 * y) % 5)] = s ? rol64(v, s) : v;
                }
            for (int y = 0; y < 5; ++y)
                for (int x = 0; x < 5; ++x) a[x + 5 * y] = b[x + 5 * y] ^ (~b[(x + 1) % 5 + 5 * y] & b[(x + 2) % 5 + 5 * y]);
            a[0] ^= RC[r];
        }
    }
    memcpy(A, a, 200);
}

int main() {
    const size_t nb = 1 << 20;  // 136 MiB
    std::vector<unsigned char> m(nb * 136);
    uint64_t s = 88172645463325252ULL;
    for (size_t i = 0; i < m.size(); i += 8) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; memcpy(&m[i], &s, 8); }
    uint64_t st1[25] = {0}, st2[25] = {0};
    absorb136_bmi(st1, m.data(), 1000);
    absorb136_lanes(st2, m.data(), 1000);
    uint64_t st3[25] = {0}, st4[25] = {0};
    absorb136_rows_d(st3, m.data(), 1000); absorb136_rows_fold(st4, m.data(), 1000);
    printf("{\"equal\": %s", memcmp(st1, st2, 200) == 0 && memcmp(st1, st3, 200) == 0 && memcmp(st1, st4, 200) == 0 ? "true" : "false");
    typedef void (*fn_t)(uint64_t*, const unsigned char*, size_t);
    const fn_t fns[4] = {absorb136_bmi, absorb136_lanes, absorb136_rows_d, absorb136_rows_fold};
    const char* names[4] = {"loop64", "lanes", "rows_d", "rows_fold"};
    for (int which = 0; which < 4; ++which) {
        double best = 1e300;
        for (int rep = 0; rep < 3; ++rep) {
            uint64_t st[25] = {0};
            auto t0 = std::chrono::steady_clock::now();
            fns[which](st, m.data(), nb);
            double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (dt < best) best = dt;
        }
        printf(", \"%s_GBps\": %.3f, \"%s_ns_per_perm\": %.1f", names[which], m.size() / best / 1e9, names[which], best / nb * 1e9);
    }
    printf("}\n");
    return 0;
}
