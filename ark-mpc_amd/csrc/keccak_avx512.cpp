// keccak_avx512.cpp -- AVX-512 Keccak-f[1600] for the host side of the hash commitment (H1, commitment.rs:30-43 / :63-89).
// The sponge is sequential by the reference's definition, so config 5 (2^24 shares) is bounded by ONE core absorbing
// 512 MiB per commitment; this is that core's inner loop.  Host code, x86-64 only, selected at run time (the library is
// built on a machine that may lack AVX-512; the MI355X hosts are EPYC 9005 parts with a full-width AVX-512 datapath).
//
// State layout: five zmm registers P0..P4, P_y holding lanes A[0..4][y] in 64-bit slots 0..4 (slots 5..7 carry garbage
// that never reaches slots 0..4).  One round =
//   theta : C = P0^P1^P2^P3^P4; D = perm(C, x-1) ^ rol(perm(C, x+1), 1); P_y ^= D          (vpternlogq 0x96, vpermq, vprolq)
//   rho   : Q_y = rolv(P_y, r[.][y])                                                         (vprolvq)
//   pi    : B[X][Y] = Q_X[(X + 3Y) mod 5]: one lane permutation per register, after which    (vpermq)
//   chi   : T_X = Q'_X ^ (~Q'_{X+1} & Q'_{X+2}) is lane-wise ACROSS registers                 (vpternlogq 0xD2)
//   iota  : T_0[0] ^= RC
//   and a 5x5 transpose T_x[y] -> P_y[x] (vpermt2q) puts the state back in plane form: 40 vector ops per round.
#include <cstddef>
#include <cstdint>
#include <cstring>

#if defined(__x86_64__)
#include <immintrin.h>
#define ARK_T512 __attribute__((target("avx512f,avx512vl")))

namespace {
const uint64_t RC[24] = {
    0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL, 0x000000000000808bULL,
    0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL, 0x000000000000008aULL, 0x0000000000000088ULL,
    0x0000000080008009ULL, 0x000000008000000aULL, 0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL,
    0x8000000000008003ULL, 0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
    0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
}

extern "C" __attribute__((visibility("hidden"))) int arkmpc_cpu_has_avx512(void) {
    return __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512vl");
}

// absorb `nblocks` full 136-byte blocks (SHA3-256 rate) into the 25-lane state
extern "C" __attribute__((visibility("hidden"))) ARK_T512 void arkmpc_keccak_absorb136_avx512(uint64_t st[25], const unsigned char* data, size_t nblocks) {
    const __mmask8 m5 = 0x1F, m2 = 0x03;
    __m512i P0 = _mm512_maskz_loadu_epi64(m5, st), P1 = _mm512_maskz_loadu_epi64(m5, st + 5), P2 = _mm512_maskz_loadu_epi64(m5, st + 10),
            P3 = _mm512_maskz_loadu_epi64(m5, st + 15), P4 = _mm512_maskz_loadu_epi64(m5, st + 20);
    const __m512i prev = _mm512_setr_epi64(4, 0, 1, 2, 3, 5, 6, 7), next = _mm512_setr_epi64(1, 2, 3, 4, 0, 5, 6, 7);
    const __m512i R0 = _mm512_setr_epi64(0, 1, 62, 28, 27, 0, 0, 0), R1 = _mm512_setr_epi64(36, 44, 6, 55, 20, 0, 0, 0),
                  R2 = _mm512_setr_epi64(3, 10, 43, 25, 39, 0, 0, 0), R3 = _mm512_setr_epi64(41, 45, 15, 21, 8, 0, 0, 0),
                  R4 = _mm512_setr_epi64(18, 2, 61, 56, 14, 0, 0, 0);
    // pi: register X -> slot Y takes old slot (X + 3Y) mod 5
    const __m512i I0 = _mm512_setr_epi64(0, 3, 1, 4, 2, 5, 6, 7), I1 = _mm512_setr_epi64(1, 4, 2, 0, 3, 5, 6, 7),
                  I2 = _mm512_setr_epi64(2, 0, 3, 1, 4, 5, 6, 7), I3 = _mm512_setr_epi64(3, 1, 4, 2, 0, 5, 6, 7),
                  I4 = _mm512_setr_epi64(4, 2, 0, 3, 1, 5, 6, 7);
    // transpose helpers
    const __m512i ia = _mm512_setr_epi64(0, 8, 1, 9, 2, 10, 3, 11), ib = _mm512_setr_epi64(4, 12, 4, 12, 4, 12, 4, 12);
    const __m512i j0 = _mm512_setr_epi64(0, 1, 8, 9, 0, 0, 0, 0), j1 = _mm512_setr_epi64(2, 3, 10, 11, 0, 0, 0, 0),
                  j2 = _mm512_setr_epi64(4, 5, 12, 13, 0, 0, 0, 0), j3 = _mm512_setr_epi64(6, 7, 14, 15, 0, 0, 0, 0);
    const __m512i k0 = _mm512_set1_epi64(0), k1 = _mm512_set1_epi64(1), k2 = _mm512_set1_epi64(2), k3 = _mm512_set1_epi64(3), k4 = _mm512_set1_epi64(4);
    for (size_t blk = 0; blk < nblocks; ++blk, data += 136) {
        P0 = _mm512_xor_si512(P0, _mm512_maskz_loadu_epi64(m5, data));
        P1 = _mm512_xor_si512(P1, _mm512_maskz_loadu_epi64(m5, data + 40));
        P2 = _mm512_xor_si512(P2, _mm512_maskz_loadu_epi64(m5, data + 80));
        P3 = _mm512_xor_si512(P3, _mm512_maskz_loadu_epi64(m2, data + 120));
        for (int r = 0; r < 24; ++r) {
            // theta
            __m512i C = _mm512_ternarylogic_epi64(_mm512_ternarylogic_epi64(P0, P1, P2, 0x96), P3, P4, 0x96);
            __m512i D = _mm512_xor_si512(_mm512_permutexvar_epi64(prev, C), _mm512_rol_epi64(_mm512_permutexvar_epi64(next, C), 1));
            // theta (apply) + rho + pi
            __m512i Q0 = _mm512_permutexvar_epi64(I0, _mm512_rolv_epi64(_mm512_xor_si512(P0, D), R0));
            __m512i Q1 = _mm512_permutexvar_epi64(I1, _mm512_rolv_epi64(_mm512_xor_si512(P1, D), R1));
            __m512i Q2 = _mm512_permutexvar_epi64(I2, _mm512_rolv_epi64(_mm512_xor_si512(P2, D), R2));
            __m512i Q3 = _mm512_permutexvar_epi64(I3, _mm512_rolv_epi64(_mm512_xor_si512(P3, D), R3));
            __m512i Q4 = _mm512_permutexvar_epi64(I4, _mm512_rolv_epi64(_mm512_xor_si512(P4, D), R4));
            // chi across registers (slot Y of T_X = new A[X][Y]) + iota
            __m512i T0 = _mm512_ternarylogic_epi64(Q0, Q1, Q2, 0xD2);
            __m512i T1 = _mm512_ternarylogic_epi64(Q1, Q2, Q3, 0xD2);
            __m512i T2 = _mm512_ternarylogic_epi64(Q2, Q3, Q4, 0xD2);
            __m512i T3 = _mm512_ternarylogic_epi64(Q3, Q4, Q0, 0xD2);
            __m512i T4 = _mm512_ternarylogic_epi64(Q4, Q0, Q1, 0xD2);
            T0 = _mm512_mask_xor_epi64(T0, 0x01, T0, _mm512_set1_epi64((long long)RC[r]));
            // transpose back to planes: P_y[x] = T_x[y]
            const __m512i A01 = _mm512_permutex2var_epi64(T0, ia, T1), A23 = _mm512_permutex2var_epi64(T2, ia, T3);
            const __m512i B01 = _mm512_permutex2var_epi64(T0, ib, T1), B23 = _mm512_permutex2var_epi64(T2, ib, T3);
            P0 = _mm512_mask_permutexvar_epi64(_mm512_permutex2var_epi64(A01, j0, A23), 0x10, k0, T4);
            P1 = _mm512_mask_permutexvar_epi64(_mm512_permutex2var_epi64(A01, j1, A23), 0x10, k1, T4);
            P2 = _mm512_mask_permutexvar_epi64(_mm512_permutex2var_epi64(A01, j2, A23), 0x10, k2, T4);
            P3 = _mm512_mask_permutexvar_epi64(_mm512_permutex2var_epi64(A01, j3, A23), 0x10, k3, T4);
            P4 = _mm512_mask_permutexvar_epi64(_mm512_permutex2var_epi64(B01, j0, B23), 0x10, k4, T4);
        }
    }
    _mm512_mask_storeu_epi64(st, m5, P0); _mm512_mask_storeu_epi64(st + 5, m5, P1); _mm512_mask_storeu_epi64(st + 10, m5, P2);
    _mm512_mask_storeu_epi64(st + 15, m5, P3); _mm512_mask_storeu_epi64(st + 20, m5, P4);
}


// ---- one lane per vector register (round 5) -----------------------------------------------------------------------------
// The plane form above pays 17 lane permutations per round for its 5-wide registers, and they sit on the round's critical path.
// With AVX-512VL the 25 lanes fit in 25 of the 32 xmm registers as plain 64-bit values and a round is 90 operations, none of them
// a permutation: the column parity is two three-input XORs (vpternlogq 0x96), theta's apply is one more (a ^ c[x-1] ^ rol(c[x+1], 1)),
// every rho rotation is one vprolq (any count), chi is one vpternlogq 0xD2 per lane, pi is register naming.  Two rounds per loop
// iteration so the compiler does not have to move lanes back at the loop edge.
namespace { typedef __m128i V; }
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

extern "C" __attribute__((visibility("hidden"))) ARK_T512 void arkmpc_keccak_absorb136_lanes(uint64_t st[25], const unsigned char* data, size_t nblocks) {
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

// The same round scheduled ROW BY ROW (round 5, late): theta's D in five registers, then for each row of the new state its five B values and
// the row's chi at once -- five B values live at a time instead of twenty-five, so the 25 lanes + C + D fit the 32 registers without spills.
// 95 operations instead of 90 (the three-input fold of theta's apply is given up for the shorter live ranges) and still faster on Zen 5:
// 0.82 GB/s built with g++, 0.87 with ROCm's clang++ against 0.77 for the form above (probes/keccak_lanes_probe.cpp).
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
extern "C" __attribute__((visibility("hidden"))) ARK_T512 void arkmpc_keccak_absorb136_rows(uint64_t st[25], const unsigned char* data, size_t nblocks) {
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

#undef X3
#undef CHI
#undef ROL
#undef LD
#undef ROUND
#undef ROUND2

// ---- portable 64-bit code, compiled twice: baseline x86-64 and with BMI1/BMI2 (andn, rorx) ------------------------------
namespace {
static inline __attribute__((always_inline)) uint64_t rol64(uint64_t x, int s) { return (x << s) | (x >> (64 - s)); }
static inline __attribute__((always_inline)) void absorb136_body(uint64_t* A, const unsigned char* data, size_t nblocks) {
    uint64_t a00 = A[0], a01 = A[1], a02 = A[2], a03 = A[3], a04 = A[4], a05 = A[5], a06 = A[6], a07 = A[7], a08 = A[8], a09 = A[9],
             a10 = A[10], a11 = A[11], a12 = A[12], a13 = A[13], a14 = A[14], a15 = A[15], a16 = A[16], a17 = A[17], a18 = A[18],
             a19 = A[19], a20 = A[20], a21 = A[21], a22 = A[22], a23 = A[23], a24 = A[24];
    for (size_t blk = 0; blk < nblocks; ++blk, data += 136) {
        uint64_t w[17];
        memcpy(w, data, 136);
        a00 ^= w[0]; a01 ^= w[1]; a02 ^= w[2]; a03 ^= w[3]; a04 ^= w[4]; a05 ^= w[5]; a06 ^= w[6]; a07 ^= w[7]; a08 ^= w[8];
        a09 ^= w[9]; a10 ^= w[10]; a11 ^= w[11]; a12 ^= w[12]; a13 ^= w[13]; a14 ^= w[14]; a15 ^= w[15]; a16 ^= w[16];
        for (int r = 0; r < 24; ++r) {
            uint64_t c0 = a00 ^ a05 ^ a10 ^ a15 ^ a20, c1 = a01 ^ a06 ^ a11 ^ a16 ^ a21, c2 = a02 ^ a07 ^ a12 ^ a17 ^ a22,
                     c3 = a03 ^ a08 ^ a13 ^ a18 ^ a23, c4 = a04 ^ a09 ^ a14 ^ a19 ^ a24;
            uint64_t d0 = c4 ^ rol64(c1, 1), d1 = c0 ^ rol64(c2, 1), d2 = c1 ^ rol64(c3, 1), d3 = c2 ^ rol64(c4, 1), d4 = c3 ^ rol64(c0, 1);
            uint64_t b00 = a00 ^ d0, b10 = rol64(a01 ^ d1, 1), b20 = rol64(a02 ^ d2, 62), b05 = rol64(a03 ^ d3, 28), b15 = rol64(a04 ^ d4, 27);
            uint64_t b16 = rol64(a05 ^ d0, 36), b01 = rol64(a06 ^ d1, 44), b11 = rol64(a07 ^ d2, 6), b21 = rol64(a08 ^ d3, 55), b06 = rol64(a09 ^ d4, 20);
            uint64_t b07 = rol64(a10 ^ d0, 3), b17 = rol64(a11 ^ d1, 10), b02 = rol64(a12 ^ d2, 43), b12 = rol64(a13 ^ d3, 25), b22 = rol64(a14 ^ d4, 39);
            uint64_t b23 = rol64(a15 ^ d0, 41), b08 = rol64(a16 ^ d1, 45), b18 = rol64(a17 ^ d2, 15), b03 = rol64(a18 ^ d3, 21), b13 = rol64(a19 ^ d4, 8);
            uint64_t b14 = rol64(a20 ^ d0, 18), b24 = rol64(a21 ^ d1, 2), b09 = rol64(a22 ^ d2, 61), b19 = rol64(a23 ^ d3, 56), b04 = rol64(a24 ^ d4, 14);
            a00 = b00 ^ (~b01 & b02) ^ RC[r]; a01 = b01 ^ (~b02 & b03); a02 = b02 ^ (~b03 & b04); a03 = b03 ^ (~b04 & b00); a04 = b04 ^ (~b00 & b01);
            a05 = b05 ^ (~b06 & b07); a06 = b06 ^ (~b07 & b08); a07 = b07 ^ (~b08 & b09); a08 = b08 ^ (~b09 & b05); a09 = b09 ^ (~b05 & b06);
            a10 = b10 ^ (~b11 & b12); a11 = b11 ^ (~b12 & b13); a12 = b12 ^ (~b13 & b14); a13 = b13 ^ (~b14 & b10); a14 = b14 ^ (~b10 & b11);
            a15 = b15 ^ (~b16 & b17); a16 = b16 ^ (~b17 & b18); a17 = b17 ^ (~b18 & b19); a18 = b18 ^ (~b19 & b15); a19 = b19 ^ (~b15 & b16);
            a20 = b20 ^ (~b21 & b22); a21 = b21 ^ (~b22 & b23); a22 = b22 ^ (~b23 & b24); a23 = b23 ^ (~b24 & b20); a24 = b24 ^ (~b20 & b21);
        }
    }
    A[0] = a00; A[1] = a01; A[2] = a02; A[3] = a03; A[4] = a04; A[5] = a05; A[6] = a06; A[7] = a07; A[8] = a08; A[9] = a09;
    A[10] = a10; A[11] = a11; A[12] = a12; A[13] = a13; A[14] = a14; A[15] = a15; A[16] = a16; A[17] = a17; A[18] = a18;
    A[19] = a19; A[20] = a20; A[21] = a21; A[22] = a22; A[23] = a23; A[24] = a24;
}
}  // namespace
extern "C" __attribute__((visibility("hidden"))) void arkmpc_keccak_absorb136_scalar(uint64_t st[25], const unsigned char* data, size_t nblocks) { absorb136_body(st, data, nblocks); }
extern "C" __attribute__((visibility("hidden"))) __attribute__((target("bmi,bmi2"))) void arkmpc_keccak_absorb136_bmi(uint64_t st[25], const unsigned char* data, size_t nblocks) {
    absorb136_body(st, data, nblocks);
}
extern "C" __attribute__((visibility("hidden"))) int arkmpc_cpu_has_bmi(void) { return __builtin_cpu_supports("bmi") && __builtin_cpu_supports("bmi2"); }
#else
extern "C" __attribute__((visibility("hidden"))) int arkmpc_cpu_has_bmi(void) { return 0; }
extern "C" __attribute__((visibility("hidden"))) void arkmpc_keccak_absorb136_scalar(uint64_t*, const unsigned char*, size_t) {}
extern "C" __attribute__((visibility("hidden"))) void arkmpc_keccak_absorb136_bmi(uint64_t*, const unsigned char*, size_t) {}
extern "C" __attribute__((visibility("hidden"))) int arkmpc_cpu_has_avx512(void) { return 0; }
extern "C" __attribute__((visibility("hidden"))) void arkmpc_keccak_absorb136_avx512(uint64_t*, const unsigned char*, size_t) {}
extern "C" __attribute__((visibility("hidden"))) void arkmpc_keccak_absorb136_lanes(uint64_t*, const unsigned char*, size_t) {}
extern "C" __attribute__((visibility("hidden"))) void arkmpc_keccak_absorb136_rows(uint64_t*, const unsigned char*, size_t) {}
#endif
