// arkmpc_curve.hip -- HIP kernels + C ABI for the curve side of the hot path (SURVEY.md section 8a
// rows a15-a19): BN254 G1 point add / sub / neg / scalar-mul (K7, K8) and the PointShare batch ops
// behind AuthenticatedPointResult (online-phase/src/algebra/curve/{curve,share,authenticated_curve}.rs).
//
// Points cross the ABI as ark-ec short-Weierstrass `Projective{x,y,z}` = Jacobian coordinates over
// Fq in Montgomery form, identity = (1,1,0).  The group law is representation-free, so results are
// compared with the reference on AFFINE coordinates (what arkworks' PartialEq and the wire format
// `serialize_compressed` see, curve.rs:103-108); the Jacobian triple itself depends on the addition
// chain.  Formulas: add-2007-bl and dbl-2009-l for a = 0 (y^2 = x^3 + 3).
//
// These kernels are integer-ALU bound (a scalar-mul is ~6k Fq multiplications against 128 B of
// traffic), one thread per point.
#include "arkmpc_internal.hpp"
#include "fp_asm.hpp"
#include <cstdlib>
#include <cstring>

// Fq multiplication used by the point formulas: the hand-scheduled block fe_mul_fast (298 instructions, 2 wait states;
// tools/gen_asm_kernels.py) unless -DARKMPC_EC_CPP selects the plain C++ core.  Measured on config 4: 12.2 ms vs 13.8 ms.
// The block's 35 fixed temporaries sit in caller-saved VGPR blocks only -- with callee-saved registers among them the
// __noinline__ point functions had to spill/restore around every call and the block was SLOWER (15.3 ms).
//
// Lazy range: inside the point formulas a coordinate lives in [0, 2q), not [0, q).  That is the multiplier block's natural
// output range -- for inputs below 2q the product is below 4q^2 and (4q^2 + R q) / R < 2q because 4q < R = 2^256 (q < 2^254) --
// so the conditional subtraction after EVERY multiplication (26 of 332 VALU instructions) disappears; add / sub / neg work
// modulo 2q at the same cost as before, zero tests accept 0 and q, and values are brought below q where they leave the
// formulas (g1_store, affine / byte conversions, comparisons of magnitudes).  Memory always holds canonical values.
constexpr int FQ = F_BN254_FQ;
constexpr int FR = F_BN254_FR;
#ifndef ARKMPC_EC_CPP
struct Fq2 {       // limbs of 2q
    static constexpr __host__ __device__ u32 L(int i) {
        return (u32)(((u64)FieldParams<F_BN254_FQ>::P(i) << 1) | (i ? (FieldParams<F_BN254_FQ>::P(i - 1) >> 31) : 0u));
    }
};
__device__ __forceinline__ Fe fq_add_lz(const Fe& a, const Fe& b) {          // (a + b) mod 2q; a + b < 4q < 2^256
    u32 s[8], d[8], c = 0, co, br = 0, bo;
#pragma unroll
    for (int i = 0; i < 8; ++i) { s[i] = __builtin_addc(a.v[i], b.v[i], c, &co); c = co; }
#pragma unroll
    for (int i = 0; i < 8; ++i) { d[i] = __builtin_subc(s[i], Fq2::L(i), br, &bo); br = bo; }
    Fe r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.v[i] = br ? s[i] : d[i];
    return r;
}
__device__ __forceinline__ Fe fq_sub_lz(const Fe& a, const Fe& b) {          // a - b (+ 2q on borrow)
    u32 d[8], br = 0, bo, c = 0, co;
#pragma unroll
    for (int i = 0; i < 8; ++i) { d[i] = __builtin_subc(a.v[i], b.v[i], br, &bo); br = bo; }
    const u32 mask = 0u - br;
    Fe r;
#pragma unroll
    for (int i = 0; i < 8; ++i) { r.v[i] = __builtin_addc(d[i], Fq2::L(i) & mask, c, &co); c = co; }
    return r;
}
__device__ __forceinline__ Fe fq_neg_lz(const Fe& a) {                       // 2q - a, and 0 stays 0
    u32 d[8], br = 0, bo, nz = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) { d[i] = __builtin_subc(Fq2::L(i), a.v[i], br, &bo); br = bo; nz |= a.v[i]; }
    Fe r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.v[i] = nz ? d[i] : 0u;
    return r;
}
__device__ __forceinline__ Fe fq_canon(const Fe& a) {                        // [0, 2q) -> [0, q)
    u32 d[8], br = 0, bo;
#pragma unroll
    for (int i = 0; i < 8; ++i) { d[i] = __builtin_subc(a.v[i], FieldParams<F_BN254_FQ>::P(i), br, &bo); br = bo; }
    Fe r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.v[i] = br ? a.v[i] : d[i];
    return r;
}
__device__ __forceinline__ bool fq_is_zero_lz(const Fe& a) {                 // 0 or q
    u32 z = 0, e = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) { z |= a.v[i]; e |= a.v[i] ^ FieldParams<F_BN254_FQ>::P(i); }
    return z == 0 || e == 0;
}
#define FQ_MUL(a, b) fe_mont_mul_asm<F_BN254_FQ>(a, b)
#define FQ_ADD(a, b) fq_add_lz(a, b)
#define FQ_SUB(a, b) fq_sub_lz(a, b)
#define FQ_NEG(a) fq_neg_lz(a)
#define FQ_CANON(a) fq_canon(a)
#define FQ_ISZERO(a) fq_is_zero_lz(a)
#else
#define FQ_MUL(a, b) fe_mul<F_BN254_FQ>(a, b)
#define FQ_ADD(a, b) fe_add<F_BN254_FQ>(a, b)
#define FQ_SUB(a, b) fe_sub<F_BN254_FQ>(a, b)
#define FQ_NEG(a) fe_neg<F_BN254_FQ>(a)
#define FQ_CANON(a) (a)
#define FQ_ISZERO(a) fe_is_zero(a)
#endif
#define FQ_SQR(a) FQ_MUL(a, a)
#define FQ_DBL(a) FQ_ADD(a, a)
#define FQ_EQ(a, b) FQ_ISZERO(FQ_SUB(a, b))

#define TPB_EC 128

struct G1 {
    Fe x, y, z;
};

__device__ __forceinline__ G1 g1_load(const u64* p) {
    G1 r;
    r.x = fe_load(p); r.y = fe_load(p + 4); r.z = fe_load(p + 8);
    return r;
}
// stores the canonical identity (1,1,0) for any z == 0 value
__device__ __forceinline__ void g1_store(u64* p, const G1& a) {
    const bool inf = FQ_ISZERO(a.z);
    const Fe one = fe_one<FQ>();
    fe_store(p, fe_select(inf, one, FQ_CANON(a.x)));
    fe_store(p + 4, fe_select(inf, one, FQ_CANON(a.y)));
    fe_store(p + 8, inf ? fe_zero<FQ>() : FQ_CANON(a.z));
}
__device__ __forceinline__ G1 g1_identity() {
    G1 r;
    r.x = fe_one<FQ>(); r.y = fe_one<FQ>(); r.z = fe_zero<FQ>();
    return r;
}
__device__ __forceinline__ G1 g1_generator() {  // (1, 2, 1)
    G1 r;
    r.x = fe_one<FQ>(); r.y = FQ_DBL(fe_one<FQ>()); r.z = fe_one<FQ>();
    return r;
}
__device__ __forceinline__ G1 g1_select(bool c, const G1& a, const G1& b) {
    G1 r;
    r.x = fe_select(c, a.x, b.x); r.y = fe_select(c, a.y, b.y); r.z = fe_select(c, a.z, b.z);
    return r;
}
__device__ __forceinline__ G1 g1_neg(const G1& a) {
    G1 r = a;
    r.y = FQ_NEG(a.y);
    return r;
}
// dbl-2009-l (a = 0).  z = 0 in -> z = 0 out, so the identity needs no branch.
__device__ __noinline__ G1 g1_double(G1 p) {
    Fe A = FQ_SQR(p.x), B = FQ_SQR(p.y), C = FQ_SQR(B);
    Fe t = FQ_SQR(FQ_ADD(p.x, B));
    Fe D = FQ_DBL(FQ_SUB(FQ_SUB(t, A), C));
    Fe E = FQ_ADD(FQ_DBL(A), A);
    Fe Fq_ = FQ_SQR(E);
    G1 r;
    r.x = FQ_SUB(Fq_, FQ_DBL(D));
    Fe C8 = FQ_DBL(FQ_DBL(FQ_DBL(C)));
    r.y = FQ_SUB(FQ_MUL(E, FQ_SUB(D, r.x)), C8);
    r.z = FQ_DBL(FQ_MUL(p.y, p.z));
    return r;
}
// add-2007-bl with the exceptional cases of the group law handled explicitly
// (identity operands, P + P, P + (-P)), as ark-ec's `Projective += Projective` does.
__device__ __noinline__ G1 g1_add(G1 p, G1 q) {
    const bool pinf = FQ_ISZERO(p.z), qinf = FQ_ISZERO(q.z);
    Fe Z1Z1 = FQ_SQR(p.z), Z2Z2 = FQ_SQR(q.z);
    Fe U1 = FQ_MUL(p.x, Z2Z2), U2 = FQ_MUL(q.x, Z1Z1);
    Fe S1 = FQ_MUL(FQ_MUL(p.y, q.z), Z2Z2), S2 = FQ_MUL(FQ_MUL(q.y, p.z), Z1Z1);
    Fe H = FQ_SUB(U2, U1);
    Fe rr = FQ_DBL(FQ_SUB(S2, S1));
    G1 out;
    if (!pinf && !qinf && FQ_ISZERO(H)) {  // same x: doubling or inverse points (rare, divergent)
        if (FQ_ISZERO(rr)) return g1_double(p);
        return g1_identity();
    }
    Fe I = FQ_SQR(FQ_DBL(H));
    Fe J = FQ_MUL(H, I);
    Fe V = FQ_MUL(U1, I);
    out.x = FQ_SUB(FQ_SUB(FQ_SQR(rr), J), FQ_DBL(V));
    out.y = FQ_SUB(FQ_MUL(rr, FQ_SUB(V, out.x)), FQ_DBL(FQ_MUL(S1, J)));
    Fe zz = FQ_SUB(FQ_SUB(FQ_SQR(FQ_ADD(p.z, q.z)), Z1Z1), Z2Z2);
    out.z = FQ_MUL(zz, H);
    out = g1_select(qinf, p, out);
    out = g1_select(pinf, q, out);
    return out;
}
// mixed addition madd-2007-bl: Jacobian p + affine (x2, y2), never called with an affine identity
__device__ __noinline__ G1 g1_madd(G1 p, Fe x2, Fe y2) {
    const bool pinf = FQ_ISZERO(p.z);
    Fe Z1Z1 = FQ_SQR(p.z);
    Fe U2 = FQ_MUL(x2, Z1Z1);
    Fe S2 = FQ_MUL(FQ_MUL(y2, p.z), Z1Z1);
    Fe H = FQ_SUB(U2, p.x);
    Fe rr = FQ_DBL(FQ_SUB(S2, p.y));
    G1 q;
    q.x = x2; q.y = y2; q.z = fe_one<FQ>();
    if (!pinf && FQ_ISZERO(H)) {          // same x: the bucket holds this point already (double) or its negative
        if (FQ_ISZERO(rr)) return g1_double(q);
        return g1_identity();
    }
    Fe HH = FQ_SQR(H);
    Fe I = FQ_DBL(FQ_DBL(HH));
    Fe J = FQ_MUL(H, I);
    Fe V = FQ_MUL(p.x, I);
    G1 out;
    out.x = FQ_SUB(FQ_SUB(FQ_SQR(rr), J), FQ_DBL(V));
    out.y = FQ_SUB(FQ_MUL(rr, FQ_SUB(V, out.x)), FQ_DBL(FQ_MUL(p.y, J)));
    out.z = FQ_SUB(FQ_SUB(FQ_SQR(FQ_ADD(p.z, H)), Z1Z1), HH);
    return g1_select(pinf, q, out);
}
// window table T[k-1] = k*P for k = 1..15 in the HBM workspace: even entries by doubling (7 Fq-mults) the half entry read
// back from the table, odd ones by adding P (16): 7 dbl + 7 add instead of 1 dbl + 13 add
__device__ __forceinline__ void g1_build_table(const G1& p, u64* tab, size_t tid, size_t nthreads, int entries = 15) {
    g1_store(tab + ((size_t)0 * nthreads + tid) * 12, p);
    G1 prev = p;                       // T[k-1]
    for (int k = 2; k <= entries; ++k) {
        G1 t;
        if (k & 1) t = g1_add(prev, p);                                                   // odd: T[k-1] + P
        else t = g1_double(g1_load(tab + ((size_t)(k / 2 - 1) * nthreads + tid) * 12));     // even: 2 * T[k/2]
        g1_store(tab + ((size_t)(k - 1) * nthreads + tid) * 12, t);
        prev = t;
    }
}
// [s]P, s given in Montgomery form over Fr (curve.rs:403-409).  Fixed 4-bit windows, MSB first: per window four
// doublings and at most one addition of a table entry k*P (k = 1..15).  The table lives in an HBM workspace
// (15 x 96 B per thread, entry-major so a wave's accesses to one entry are contiguous): 160 KiB of LDS could only
// hold it at one wave per CU, and this kernel is integer-ALU bound (~3.0 k Fq multiplications per scalar-mul against
// ~6 KiB of table reads).  Control flow is wave-uniform; the addition runs when any lane has a non-zero digit and is
// merged per lane with a select.
__device__ __forceinline__ G1 g1_scalar_mul_w4(const G1& p, const Fe& s_mont, u64* tab, size_t tid, size_t nthreads) {
    const Fe s = fe_to_canonical<FR>(s_mont);
    g1_build_table(p, tab, tid, nthreads);
    G1 acc = g1_identity();
    for (int limb = 7; limb >= 0; --limb) {
        const u32 w = s.v[limb];
        for (int nib = 7; nib >= 0; --nib) {
            acc = g1_double(acc);
            acc = g1_double(acc);
            acc = g1_double(acc);
            acc = g1_double(acc);
            const u32 d = (w >> (4 * nib)) & 15u;
            if (__any(d != 0)) {
                const u32 k = d ? d - 1 : 0;
                G1 q = g1_load(tab + ((size_t)k * nthreads + tid) * 12);
                G1 sum = g1_add(acc, q);
                acc = g1_select(d != 0, sum, acc);
            }
        }
    }
    return acc;
}
// ---------------------------------------------------------------------------------------------
// GLV: phi(x, y) = (beta x, y) = [lambda](x, y) on BN254 G1, so [k]P = [k1]P + [k2]phi(P) with |k1|, |k2| < 2^128
// (tools/gen_glv_consts.py derives the lattice basis and models the exact integer steps below).  One shared run of
// doublings serves both half-length scalars: 33 windows x (4 dbl + <= 2 add) ~ 2.2 k Fq multiplications instead of
// ~3.0 k.  The phi-table is never stored: phi(T[d]) costs one multiplication by beta when the entry is loaded.
// k1 + k2*lambda == k (mod r) holds for any integers c1, c2, so the floor approximations affect only magnitudes.
// ---------------------------------------------------------------------------------------------
#include "glv_consts.inc"

// out[NA+NB] = a[NA] * b[NB] (unsigned, little-endian u32 limbs)
template <int NA, int NB> __device__ __forceinline__ void bn_mul(const u32 (&a)[NA], const u32 (&b)[NB], u32 (&out)[NA + NB]) {
#pragma unroll
    for (int i = 0; i < NA + NB; ++i) out[i] = 0;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        u64 c = 0;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            u64 t = (u64)a[i] * b[j] + out[i + j] + c;
            out[i + j] = (u32)t; c = t >> 32;
        }
        out[NA + j] = (u32)c;
    }
}
// acc[9] (two's complement) += sign * v[9]
__device__ __forceinline__ void bn_addsub9(u32 (&acc)[9], const u32 (&v)[9], bool subtract) {
    u32 c = subtract ? 1u : 0u, co;
#pragma unroll
    for (int i = 0; i < 9; ++i) { acc[i] = __builtin_addc(acc[i], subtract ? ~v[i] : v[i], c, &co); c = co; }
}
struct GlvHalf { u32 mag[5]; bool neg; };
__device__ __forceinline__ GlvHalf glv_abs(u32 (&acc)[9]) {
    GlvHalf h;
    h.neg = (acc[8] >> 31) != 0;
    u32 c = h.neg ? 1u : 0u, co;
#pragma unroll
    for (int i = 0; i < 9; ++i) { u32 w = h.neg ? ~acc[i] : acc[i]; acc[i] = __builtin_addc(w, 0u, c, &co); c = co; }
#pragma unroll
    for (int i = 0; i < 5; ++i) h.mag[i] = acc[i];   // limbs 5..8 are zero: |k_i| < 2^132 (model: <= 127 bits)
    return h;
}
__device__ __forceinline__ void glv_decompose(const Fe& k, GlvHalf& h1, GlvHalf& h2) {
    u32 kk[8], g1[5], g2[5], p1[13], p2[13], c1[5], c2[5];
#pragma unroll
    for (int i = 0; i < 8; ++i) kk[i] = k.v[i];
#pragma unroll
    for (int i = 0; i < 5; ++i) { g1[i] = GLV_G1[i]; g2[i] = GLV_G2[i]; }
    bn_mul<8, 5>(kk, g1, p1);
    bn_mul<8, 5>(kk, g2, p2);
#pragma unroll
    for (int i = 0; i < 5; ++i) { c1[i] = p1[8 + i]; c2[i] = p2[8 + i]; }   // floor(k*g / 2^256)
    u32 a1[4], b1[4], a2[4], b2[4], t[9], acc1[9], acc2[9];
#pragma unroll
    for (int i = 0; i < 4; ++i) { a1[i] = GLV_A1_MAG[i]; b1[i] = GLV_B1_MAG[i]; a2[i] = GLV_A2_MAG[i]; b2[i] = GLV_B2_MAG[i]; }
    // k1 = k - c1*a1 - c2*a2
#pragma unroll
    for (int i = 0; i < 9; ++i) { acc1[i] = (i < 8) ? kk[i] : 0u; acc2[i] = 0u; }
    bn_mul<5, 4>(c1, a1, t); bn_addsub9(acc1, t, (GLV_C1_NEG ^ GLV_A1_NEG) == 0);
    bn_mul<5, 4>(c2, a2, t); bn_addsub9(acc1, t, (GLV_C2_NEG ^ GLV_A2_NEG) == 0);
    // k2 = -c1*b1 - c2*b2
    bn_mul<5, 4>(c1, b1, t); bn_addsub9(acc2, t, (GLV_C1_NEG ^ GLV_B1_NEG) == 0);
    bn_mul<5, 4>(c2, b2, t); bn_addsub9(acc2, t, (GLV_C2_NEG ^ GLV_B2_NEG) == 0);
    h1 = glv_abs(acc1);
    h2 = glv_abs(acc2);
}
__device__ __forceinline__ G1 g1_scalar_mul_glv(const G1& p, const Fe& s_mont, u64* tab, size_t tid, size_t nthreads) {
    const Fe s = fe_to_canonical<FR>(s_mont);
    GlvHalf h1, h2;
    glv_decompose(s, h1, h2);
    Fe beta;
#pragma unroll
    for (int i = 0; i < 8; ++i) beta.v[i] = GLV_BETA_MONT[i];
    g1_build_table(p, tab, tid, nthreads);
    G1 acc = g1_identity();
    for (int w = GLV_WINDOWS - 1; w >= 0; --w) {
        acc = g1_double(acc);
        acc = g1_double(acc);
        acc = g1_double(acc);
        acc = g1_double(acc);
        const u32 d1 = (h1.mag[w >> 3] >> (4 * (w & 7))) & 15u;
        const u32 d2 = (h2.mag[w >> 3] >> (4 * (w & 7))) & 15u;
        if (__any(d1 != 0)) {
            G1 q = g1_load(tab + ((size_t)(d1 ? d1 - 1 : 0) * nthreads + tid) * 12);
            if (h1.neg) q.y = FQ_NEG(q.y);
            G1 sum = g1_add(acc, q);
            acc = g1_select(d1 != 0, sum, acc);
        }
        if (__any(d2 != 0)) {
            G1 q = g1_load(tab + ((size_t)(d2 ? d2 - 1 : 0) * nthreads + tid) * 12);
            q.x = FQ_MUL(q.x, beta);                 // phi on Jacobian coordinates: (beta X, Y, Z)
            if (h2.neg) q.y = FQ_NEG(q.y);
            G1 sum = g1_add(acc, q);
            acc = g1_select(d2 != 0, sum, acc);
        }
    }
    return acc;
}
// Signed 5-bit windows for the two GLV halves: digits in [-15, 16] (a digit above 16 becomes d - 32 with a carry into the next
// window), so 27 windows x 2 additions replace 33 x 2 and the table grows by one entry (16 P): ~165 Fq multiplications fewer
// per scalar-mul than the unsigned 4-bit form above.  Digits are recoded LSB-first once (fully unrolled, in registers), packed
// five to a word: bits 0-4 = |d|, bit 5 = negative.
#define GLV5_WINDOWS 27        // 135 bits >= the 132 the decomposition model bounds, plus the top carry
struct GlvDigits { u32 w[6]; };
__device__ __forceinline__ GlvDigits glv_recode5(const u32 (&mag)[5]) {
    GlvDigits r;
#pragma unroll
    for (int k = 0; k < 6; ++k) r.w[k] = 0;
    u32 carry = 0;
#pragma unroll
    for (int j = 0; j < GLV5_WINDOWS; ++j) {
        const int bit = 5 * j, limb = bit >> 5, sh = bit & 31;
        u32 v = mag[limb] >> sh;
        if (sh > 27 && limb + 1 < 5) v |= mag[limb + 1] << (32 - sh);
        u32 d = (v & 31u) + carry, neg = 0;
        if (d > 16u) { d = 32u - d; neg = 1; carry = 1; } else carry = 0;
        r.w[j / 5] |= (d | (neg << 5)) << (6 * (j % 5));
    }
    return r;
}
__device__ __forceinline__ u32 glv_digit(const GlvDigits& d, int w) {
    const int word = w / 5, pos = w - 5 * word;
    u32 v = d.w[0];
#pragma unroll
    for (int k = 1; k < 6; ++k) v = (word == k) ? d.w[k] : v;
    return (v >> (6 * pos)) & 63u;
}
__device__ __forceinline__ G1 g1_scalar_mul_glv5(const G1& p, const Fe& s_mont, u64* tab, size_t tid, size_t nthreads) {
    const Fe s = fe_to_canonical<FR>(s_mont);
    GlvHalf h1, h2;
    glv_decompose(s, h1, h2);
    const GlvDigits D1 = glv_recode5(h1.mag), D2 = glv_recode5(h2.mag);
    Fe beta;
#pragma unroll
    for (int i = 0; i < 8; ++i) beta.v[i] = GLV_BETA_MONT[i];
    g1_build_table(p, tab, tid, nthreads, 16);
    G1 acc = g1_identity();
    for (int w = GLV5_WINDOWS - 1; w >= 0; --w) {
        if (w != GLV5_WINDOWS - 1) {
            acc = g1_double(acc); acc = g1_double(acc); acc = g1_double(acc); acc = g1_double(acc); acc = g1_double(acc);
        }
        const u32 e1 = glv_digit(D1, w), e2 = glv_digit(D2, w);
        const u32 m1 = e1 & 31u, m2 = e2 & 31u;
        if (__any(m1 != 0)) {
            G1 q = g1_load(tab + ((size_t)(m1 ? m1 - 1 : 0) * nthreads + tid) * 12);
            if (h1.neg != (bool)(e1 >> 5)) q.y = FQ_NEG(q.y);
            G1 sum = g1_add(acc, q);
            acc = g1_select(m1 != 0, sum, acc);
        }
        if (__any(m2 != 0)) {
            G1 q = g1_load(tab + ((size_t)(m2 ? m2 - 1 : 0) * nthreads + tid) * 12);
            q.x = FQ_MUL(q.x, beta);                 // phi on Jacobian coordinates: (beta X, Y, Z)
            if (h2.neg != (bool)(e2 >> 5)) q.y = FQ_NEG(q.y);
            G1 sum = g1_add(acc, q);
            acc = g1_select(m2 != 0, sum, acc);
        }
    }
    return acc;
}
// plain MSB-first double-and-add (used for the single uniform-key multiplications inside other kernels)
__device__ __forceinline__ G1 g1_scalar_mul(const G1& p, const Fe& s_mont) {
    const Fe s = fe_to_canonical<FR>(s_mont);
    G1 acc = g1_identity();
    for (int limb = 7; limb >= 0; --limb) {
        const u32 w = s.v[limb];
        for (int bit = 31; bit >= 0; --bit) {
            acc = g1_double(acc);
            const bool take = (w >> bit) & 1u;
            if (__any(take)) {
                G1 sum = g1_add(acc, p);
                acc = g1_select(take, sum, acc);
            }
        }
    }
    return acc;
}
// Fq inversion by Fermat (a^(p-2)); only used by the affine / byte conversions
__device__ __noinline__ Fe fq_inv(const Fe& a) {
    using P = FieldParams<FQ>;
    Fe acc = fe_one<FQ>();
    for (int limb = 7; limb >= 0; --limb) {
        u32 w = P::P(limb);
        if (limb == 0) w -= 2u;  // p - 2 (p is odd and its low limb is > 2, so no borrow)
        for (int bit = 31; bit >= 0; --bit) {
            acc = FQ_SQR(acc);
            if ((w >> bit) & 1u) acc = FQ_MUL(acc, a);
        }
    }
    return acc;
}
// z^-1 for a batch of points with Montgomery's trick: a thread owns K points strided by the thread count and spends ONE
// Fermat exponentiation on them (3 + 380/K multiplications per point instead of ~390); running products go to `pre`.
// Identities (z = 0) are skipped and get 0.  Results are in the lazy range (scratch, not ABI memory).
__global__ void __launch_bounds__(TPB_EC) k_g1_zinv(size_t n, u32 K, size_t nthreads, const u64* pts, u64* pre, u64* zinv) {
    const size_t t = (size_t)blockIdx.x * TPB_EC + threadIdx.x;
    if (t >= nthreads) return;
    Fe run = fe_one<FQ>();
    for (u32 j = 0; j < K; ++j) {
        const size_t i = (size_t)j * nthreads + t;
        if (i >= n) break;
        const Fe z = fe_load(pts + 12 * i + 8);
        if (!FQ_ISZERO(z)) run = FQ_MUL(run, z);
        fe_store(pre + 4 * i, run);
    }
    Fe inv = fq_inv(run);
    for (int j = (int)K - 1; j >= 0; --j) {
        const size_t i = (size_t)j * nthreads + t;
        if (i >= n) continue;
        const Fe z = fe_load(pts + 12 * i + 8);
        Fe o = fe_zero<FQ>();
        if (!FQ_ISZERO(z)) {
            const Fe before = (j == 0) ? fe_one<FQ>() : fe_load(pre + 4 * ((size_t)(j - 1) * nthreads + t));
            o = FQ_MUL(inv, before);
            inv = FQ_MUL(inv, z);
        }
        fe_store(zinv + 4 * i, o);
    }
}
__device__ __forceinline__ void g1_to_affine_zi(const G1& a, const Fe& zi, Fe& x, Fe& y, bool& inf) {
    inf = FQ_ISZERO(a.z);
    Fe zi2 = FQ_SQR(zi);
    x = FQ_CANON(FQ_MUL(a.x, zi2));
    y = FQ_CANON(FQ_MUL(a.y, FQ_MUL(zi2, zi)));
    if (inf) { x = fe_zero<FQ>(); y = fe_zero<FQ>(); }
}
__device__ __forceinline__ void g1_to_affine(const G1& a, Fe& x, Fe& y, bool& inf) {
    inf = FQ_ISZERO(a.z);
    Fe zi = fq_inv(a.z);  // 0 -> 0
    Fe zi2 = FQ_SQR(zi);
    x = FQ_CANON(FQ_MUL(a.x, zi2));
    y = FQ_CANON(FQ_MUL(a.y, FQ_MUL(zi2, zi)));
    if (inf) { x = fe_zero<FQ>(); y = fe_zero<FQ>(); }
}

// ---------------------------------------------------------------------------------------------
// Fixed-base multiplication by the generator (CurvePointResult / AuthenticatedPointResult::batch_mul_generator,
// authenticated_curve.rs:754-780; 4 of the scalar-muls of the Beaver point multiplication :696-708; input sharing of
// points, fabric.rs:622-649).  The base never changes, so its multiples are tabulated once per device:
// T[w][d-1] = d * 2^(12w) * G for the 22 window positions w and d = 1..2048, affine: 45 056 x 64 B = 2.75 MiB, which stays in
// each XCD's 4 MiB L2.  [s]G is then 22 SIGNED 12-bit digits, 22 table lookups and at most 22 mixed additions -- no doublings
// (round 1: 8-bit unsigned digits, 32 additions, 510 KiB).  The additions run on the hand-scheduled mixed-addition body of the
// scalar-mul loop (g1_msm_acc_asm, the bucket fold of the MSM: the digit list of a lane IS a bucket run); a chain that meets
// H = 0 only flags its lane and k_gen_mul_chain redoes it with the complete compiled addition.
// ---------------------------------------------------------------------------------------------
#include "ec_asm_kernels.inc"
#include "ec29_asm_kernels.inc"      // the same loop / table on nine 29-bit limbs (tools/gen_ec29_asm.py): the default; ARKMPC_EC_LIMBS=32 selects the kernels above
// nine 29-bit limbs (default) or eight 32-bit limbs (ARKMPC_EC_LIMBS=32) in the hand-scheduled kernels: window loop, table, MSM bucket fold, generator chain
static bool g1_limbs29() {
    static const bool on = !(getenv("ARKMPC_EC_LIMBS") && !strcmp(getenv("ARKMPC_EC_LIMBS"), "32"));
    return on;
}
#define GEN_C 12
#define GEN_WINDOWS 22                      // ceil(254 / 12); the top window holds 2 scalar bits + carry
#define GEN_ENTRIES (1u << (GEN_C - 1))     // |digit| = 1 .. 2048
__global__ void __launch_bounds__(64) k_gen_table_bases(u64* bases) {           // one thread: B_w = 2^(12w) G
    if (blockIdx.x | threadIdx.x) return;
    G1 b = g1_generator();
    for (int w = 0; w < GEN_WINDOWS; ++w) {
        g1_store(bases + 12 * w, b);
        for (int k = 0; k < GEN_C; ++k) b = g1_double(b);
    }
}
__device__ __forceinline__ Fe fq_words(const u32 (&c)[8]) {
    Fe r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.v[i] = c[i];
    return r;
}
__global__ void __launch_bounds__(TPB_EC) k_gen_table_fill(const u64* bases, u64* table, u64* table29) {     // thread (w, d): d * B_w, normalised
    const u32 t = blockIdx.x * TPB_EC + threadIdx.x;
    if (t >= GEN_WINDOWS * GEN_ENTRIES) return;
    const u32 w = t / GEN_ENTRIES, d = t % GEN_ENTRIES + 1;
    const G1 b = g1_load(bases + 12 * w);
    G1 acc = g1_identity();
    for (int bit = GEN_C - 1; bit >= 0; --bit) {
        acc = g1_double(acc);
        if ((d >> bit) & 1u) acc = g1_add(acc, b);
    }
    Fe x, y; bool inf;
    g1_to_affine(acc, x, y, inf);                        // never the identity: d * 2^(12w) < r
    fe_store(table + 8 * (size_t)t, x);
    fe_store(table + 8 * (size_t)t + 4, y);
    // the same entries for the 29-bit-limb chain (g1_msm_acc29_asm): Montgomery radix 2^261
    fe_store(table29 + 8 * (size_t)t, FQ_CANON(FQ_MUL(x, fq_words(G1_ASM29_TO29))));
    fe_store(table29 + 8 * (size_t)t + 4, FQ_CANON(FQ_MUL(y, fq_words(G1_ASM29_TO29))));
}
// signed 12-bit digits of the canonical scalar as a compacted member list: vals[GEN_WINDOWS * i + j] = table index | sign << 31 for the
// j-th NON-ZERO digit, lens[i] = how many there are
__global__ void __launch_bounds__(256) k_gen_digits(size_t n, const u64* scalars, u32 s_stride, u32 s_div, u32* vals, u32* lens) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const Fe s = fe_to_canonical<FR>(fe_load(scalars + (size_t)s_stride * (i / s_div)));
    u32 carry = 0, cnt = 0;
    for (u32 w = 0; w < GEN_WINDOWS; ++w) {
        const u32 bit = GEN_C * w, limb = bit >> 5, sh = bit & 31;
        u32 lo = 0, hi = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) { lo = (limb == (u32)k) ? s.v[k] : lo; hi = (limb + 1 == (u32)k) ? s.v[k] : hi; }
        const u64 both = ((u64)hi << 32) | lo;
        u32 d = ((u32)(both >> sh) & ((1u << GEN_C) - 1u)) + carry, neg = 0;
        if (d > GEN_ENTRIES) { d = (1u << GEN_C) - d; neg = 1; carry = 1; } else carry = 0;
        if (d) vals[(size_t)GEN_WINDOWS * i + cnt++] = (w * GEN_ENTRIES + (d - 1)) | (neg << 31);
    }
    lens[i] = cnt;
}
// the chain with the complete compiled addition: every lane (only == nullptr), or the lanes the asm kernel flagged
__global__ void __launch_bounds__(TPB_EC) k_gen_mul_chain(size_t n, const u32* vals, const u32* lens, const u64* table, u64* out, const u32* only) {
    const size_t i = (size_t)blockIdx.x * TPB_EC + threadIdx.x;
    if (i >= n) return;
    if (only && !only[i]) return;
    const u32 len = lens[i];
    G1 acc = g1_identity();
    for (u32 j = 0; j < len; ++j) {
        const u32 v = vals[(size_t)GEN_WINDOWS * i + j];
        const u64* q = table + 8 * (size_t)(v & 0x7fffffffu);
        Fe y = fe_load(q + 4);
        if (v >> 31) y = FQ_NEG(y);
        acc = g1_madd(acc, fe_load(q), y);
    }
    g1_store(out + 12 * i, acc);
}

// ---------------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------------
// K7: curve.rs:203-209 (+), :282-288 (-); NEGB selects subtraction (a + (-b)), as curve/share.rs:95-101
template <bool NEGB>
__global__ void __launch_bounds__(TPB_EC) k_g1_add(size_t n, const u64* a, const u64* b, u64* out) {
    size_t i = (size_t)blockIdx.x * TPB_EC + threadIdx.x;
    if (i >= n) return;
    G1 p = g1_load(a + 12 * i), q = g1_load(b + 12 * i);
    if (NEGB) q = g1_neg(q);
    g1_store(out + 12 * i, g1_add(p, q));
}
__global__ void __launch_bounds__(TPB_EC) k_g1_neg(size_t n, const u64* a, u64* out) {
    size_t i = (size_t)blockIdx.x * TPB_EC + threadIdx.x;
    if (i >= n) return;
    g1_store(out + 12 * i, g1_neg(g1_load(a + 12 * i)));
}
// K8: out_i = scalars[i * s_stride .. ] * points[i * p_stride ..]  (strides in u64 units; p_stride = 0 with
// points == nullptr means the generator).  Covers CurvePoint*Scalar (curve.rs:403-409, :459-479),
// PointShare*Scalar (curve/share.rs:108-114: two launches' worth, index i -> element i/2, scalar i/2),
// ScalarShare*CurvePoint (scalar/share.rs:135-141) and ScalarShare*generator (authenticated_curve.rs:754-780).
template <int GLV>      // 2 = GLV + signed 5-bit windows, 1 = GLV + unsigned 4-bit windows, 0 = plain 4-bit windows (one kernel each: separate register allocation)
__global__ void __launch_bounds__(TPB_EC) k_g1_scalar_mul(size_t n, const u64* points, u32 p_stride, u32 p_div,
                                                           const u64* scalars, u32 s_stride, u32 s_div, u64* out, u64* table_ws) {
    size_t i = (size_t)blockIdx.x * TPB_EC + threadIdx.x;
    if (i >= n) return;
    G1 p = points ? g1_load(points + (size_t)p_stride * (i / p_div)) : g1_generator();
    Fe s = fe_load(scalars + (size_t)s_stride * (i / s_div));
    G1 r;
    if (GLV == 2) r = g1_scalar_mul_glv5(p, s, table_ws, i, n);
    else if (GLV == 1) r = g1_scalar_mul_glv(p, s, table_ws, i, n);
    else r = g1_scalar_mul_w4(p, s, table_ws, i, n);
    g1_store(out + 12 * i, r);
}
// ---------------------------------------------------------------------------------------------
// K8, hand-scheduled form: prep -> window loop (one asm stream, ec_asm_kernels.inc / tools/gen_ec_asm.py) -> finish.
//   prep    per scalar-mul: GLV split + signed 5-bit recoding -> 55 digit records (27 windows x 2 halves + the final step);
//           the 16 multiples of P, rescaled to ONE common Z (Zc = product of their z's: x_k' = X_k c_k^2, y_k' = Y_k c_k^3 with
//           c_k = Zc / z_k from prefix / suffix products, 111 multiplications) -- on the isomorphic curve y^2 = x^3 + 3 Zc^6 they
//           are AFFINE, so the loop adds with madd-2007-bl (11 multiplications, not 16); entries 16 / 17 = the blinding point R0
//           and -(2^130 R0) mapped to that curve ((x Zc^2, y Zc^3)).
//   loop    acc = R0'; 26 x 5 doublings, 54 mixed additions of +-T[d] / +-phi(T[d]), then - 2^130 R0'.  No infinity and no
//           branches on data; lanes that hit H = 0 (P + P, P - P) only raise a flag.
//   finish  Z = Z' * Zc maps back to the curve; canonical store.  Flagged lanes (the identity as input or result, zero scalar,
//           crafted collisions with R0) are recomputed by the compiled window path, so every input gets the exact group-law result.
// Measured (config 4, 2^19 scalar-muls): see DESIGN.md section 3.
// ---------------------------------------------------------------------------------------------
#define TPB_LOOP 256
// fixed-base chain on the hand-scheduled mixed-addition body (see the generator-table section above)
__global__ void __launch_bounds__(TPB_LOOP) k_gen_mul_chain_asm(u32 n, const u32* vals, const u32* lens, const u64* table, u64* out, u32* exc) {
    const u32 i = blockIdx.x * TPB_LOOP + threadIdx.x;
    const bool valid = i < n;
    const u32 len = valid ? lens[i] : 0u;
    u32 m = len;
    for (int o = 32; o > 0; o >>= 1) { const u32 other = (u32)__shfl_xor((int)m, o); m = other > m ? other : m; }
    const u32 maxlen = __builtin_amdgcn_readfirstlane(m);
    if (!valid) return;
    if (len == 0) { g1_store(out + 12 * (size_t)i, g1_identity()); exc[i] = 0; return; }     // zero scalar
    g1_msm_acc_asm(i * (GEN_WINDOWS * 4u), len, maxlen, vals, table, out + 12 * (size_t)i, i * 4u, exc);
}
__global__ void __launch_bounds__(TPB_LOOP) k_gen_mul_chain_asm29(u32 n, const u32* vals, const u32* lens, const u64* table29, u64* out, u32* exc) {
    const u32 i = blockIdx.x * TPB_LOOP + threadIdx.x;
    const bool valid = i < n;
    const u32 len = valid ? lens[i] : 0u;
    u32 m = len;
    for (int o = 32; o > 0; o >>= 1) { const u32 other = (u32)__shfl_xor((int)m, o); m = other > m ? other : m; }
    const u32 maxlen = __builtin_amdgcn_readfirstlane(m);
    if (!valid) return;
    if (len == 0) { g1_store(out + 12 * (size_t)i, g1_identity()); exc[i] = 0; return; }     // zero scalar
    g1_msm_acc29_asm(i * (GEN_WINDOWS * 4u), len, maxlen, vals, table29, out + 12 * (size_t)i, i * 4u, exc);
}
struct G1AsmWs {
    u64* jtab;      // [16][n] Jacobian multiples (scratch of prep; window table of the fallback in finish)
    u64* tab;       // [18][n] affine entries on the isomorphic curve, 64 B each
    u32* dig;       // [55][n] step records: bits 0-4 table index, bit 5 negate, bit 6 digit non-zero
    u64* res;       // [n] raw result of the loop (X', Y', Z'), lazy range
    u64* zc;        // [n] common Z
    u32* exc0;      // [n] prep: input point is the identity
    u32* exc1;      // [n] loop: exceptional addition
};
__device__ __forceinline__ Fe fq_const(const u32 (&c)[8]) {
    Fe r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.v[i] = c[i];
    return r;
}
__global__ void __launch_bounds__(TPB_EC) k_g1_smul_prep(u32 n, const u64* points, u32 p_stride, u32 p_div, const u64* scalars, u32 s_stride,
                                                          u32 s_div, G1AsmWs ws) {
    const u32 i = blockIdx.x * TPB_EC + threadIdx.x;
    if (i >= n) return;
    const G1 p = points ? g1_load(points + (size_t)p_stride * (i / p_div)) : g1_generator();
    const Fe s = fe_to_canonical<FR>(fe_load(scalars + (size_t)s_stride * (i / s_div)));
    // ---- digit records
    GlvHalf h1, h2;
    glv_decompose(s, h1, h2);
    const GlvDigits D1 = glv_recode5(h1.mag), D2 = glv_recode5(h2.mag);
    for (int w = GLV5_WINDOWS - 1; w >= 0; --w) {
        const u32 e1 = glv_digit(D1, w), e2 = glv_digit(D2, w);
        const u32 m1 = e1 & 31u, m2 = e2 & 31u;
        const u32 r1 = (m1 ? m1 - 1 : 0u) | ((((e1 >> 5) & 1u) ^ (u32)h1.neg) << 5) | ((m1 != 0) << 6);
        const u32 r2 = (m2 ? m2 - 1 : 0u) | ((((e2 >> 5) & 1u) ^ (u32)h2.neg) << 5) | ((m2 != 0) << 6);
        const u32 step = 2u * (GLV5_WINDOWS - 1 - w);
        ws.dig[(size_t)step * n + i] = r1;
        ws.dig[(size_t)(step + 1) * n + i] = r2;
    }
    ws.dig[(size_t)(G1_ASM_STEPS - 1) * n + i] = 17u | (1u << 6);            // the final step: + (-(2^130 R0))'
    ws.exc0[i] = (FQ_ISZERO(p.z) || fe_is_zero(s)) ? 1u : 0u;                 // trivial lanes: identity in or zero scalar -> identity out
    // ---- Jacobian multiples 1..16 (as the compiled path builds them), then one common Z
    g1_build_table(p, ws.jtab, i, n, 16);
    Fe run = fe_one<FQ>();
    for (int k = 0; k < 16; ++k) {                                           // prefix products of the z's, parked in the x slots
        run = FQ_MUL(run, fe_load(ws.jtab + ((size_t)k * n + i) * 12 + 8));
        fe_store(ws.tab + ((size_t)k * n + i) * 8, FQ_CANON(run));
    }
    const Fe zc = FQ_CANON(run);
    fe_store(ws.zc + 4 * (size_t)i, zc);
    Fe suf = fe_one<FQ>();                                                   // z_{k+1} ... z_16
    for (int k = 15; k >= 0; --k) {
        const u64* jp = ws.jtab + ((size_t)k * n + i) * 12;
        const Fe pre = k ? fe_load(ws.tab + ((size_t)(k - 1) * n + i) * 8) : fe_one<FQ>();
        const Fe c = FQ_MUL(pre, suf);                                       // Zc / z_k
        const Fe c2 = FQ_SQR(c);
        u64* op = ws.tab + ((size_t)k * n + i) * 8;
        fe_store(op, FQ_CANON(FQ_MUL(fe_load(jp), c2)));
        fe_store(op + 4, FQ_CANON(FQ_MUL(fe_load(jp + 4), FQ_MUL(c2, c))));
        suf = FQ_MUL(suf, fe_load(jp + 8));
    }
    const Fe z2 = FQ_SQR(zc), z3 = FQ_MUL(z2, zc);                           // R0 and -(2^130 R0) on the isomorphic curve
    u64* o16 = ws.tab + ((size_t)16 * n + i) * 8;
    u64* o17 = ws.tab + ((size_t)17 * n + i) * 8;
    fe_store(o16, FQ_CANON(FQ_MUL(fq_const(G1_ASM_R0X), z2)));
    fe_store(o16 + 4, FQ_CANON(FQ_MUL(fq_const(G1_ASM_R0Y), z3)));
    fe_store(o17, FQ_CANON(FQ_MUL(fq_const(G1_ASM_NCX), z2)));
    fe_store(o17 + 4, FQ_CANON(FQ_MUL(fq_const(G1_ASM_NCY), z3)));
}
// prep, hand-scheduled form = two kernels: the digit records (integer work only: light, compiled) and the table (g1_smul_table_asm:
// the loop's own double / mixed-add bodies on the curve where P is affine, then the common-Z rescaling, all in one asm stream)
__global__ void __launch_bounds__(TPB_LOOP) k_g1_smul_digits(u32 n, const u64* points, u32 p_stride, u32 p_div, const u64* scalars, u32 s_stride,
                                                              u32 s_div, u32* dig, u32* exc0) {
    const u32 i = blockIdx.x * TPB_LOOP + threadIdx.x;
    if (i >= n) return;
    const Fe s = fe_to_canonical<FR>(fe_load(scalars + (size_t)s_stride * (i / s_div)));
    GlvHalf h1, h2;
    glv_decompose(s, h1, h2);
    const GlvDigits D1 = glv_recode5(h1.mag), D2 = glv_recode5(h2.mag);
    for (int w = GLV5_WINDOWS - 1; w >= 0; --w) {
        const u32 e1 = glv_digit(D1, w), e2 = glv_digit(D2, w);
        const u32 m1 = e1 & 31u, m2 = e2 & 31u;
        const u32 step = 2u * (GLV5_WINDOWS - 1 - w);
        dig[(size_t)step * n + i] = (m1 ? m1 - 1 : 0u) | ((((e1 >> 5) & 1u) ^ (u32)h1.neg) << 5) | ((m1 != 0) << 6);
        dig[(size_t)(step + 1) * n + i] = (m2 ? m2 - 1 : 0u) | ((((e2 >> 5) & 1u) ^ (u32)h2.neg) << 5) | ((m2 != 0) << 6);
    }
    dig[(size_t)(G1_ASM_STEPS - 1) * n + i] = 17u | (1u << 6);
    // trivial lanes: the identity as input or a zero scalar give the identity -- finish stores it without recomputing anything
    // (the loop would end on acc == 2^130 R0 and flag the lane; zero shares / MACs are common: a party's MAC-key share can be 0)
    exc0[i] = (fe_is_zero(fe_load(points + (size_t)p_stride * (i / p_div) + 8)) || fe_is_zero(s)) ? 1u : 0u;
}
// one table column per lane of THIS launch: the launcher passes the number of columns (= lanes / tdiv when tdiv lanes share a point)
__global__ void __launch_bounds__(TPB_LOOP) k_g1_smul_table(u32 ncol, const u64* points, u32 p_stride, u32 p_div, u64* jtab, u64* tab, u64* zc) {
    const u32 i = blockIdx.x * TPB_LOOP + threadIdx.x;
    if (i >= ncol) return;
    g1_smul_table_asm(i, p_stride * 8u * (i / p_div), ncol, points, jtab, tab, zc);
}
// 29-bit-limb forms: same arguments, same tab / zc / res formats (packed 32-bit words); jtab is private to the table kernel (27-word entries)
__global__ void __launch_bounds__(TPB_LOOP) k_g1_smul_table29(u32 ncol, const u64* points, u32 p_stride, u32 p_div, u64* jtab, u64* tab, u64* zc) {
    const u32 i = blockIdx.x * TPB_LOOP + threadIdx.x;
    if (i >= ncol) return;
    g1_smul_table29_asm(i, p_stride * 8u * (i / p_div), ncol, points, jtab, tab, zc);
}
__global__ void __launch_bounds__(TPB_LOOP) k_g1_smul_loop29(u32 n, u32 tdiv, const u64* tab, const u32* dig, u64* res, u32* exc) {
    const u32 i = blockIdx.x * TPB_LOOP + threadIdx.x;
    if (i >= n) return;
    g1_smul_loop29_asm(i, n, i / tdiv, n / tdiv, tab, dig, res, exc);
}
// tdiv lanes share one table column (ScalarShare x point: the share lane and the MAC lane multiply the same point)
__global__ void __launch_bounds__(TPB_LOOP) k_g1_smul_loop(u32 n, u32 tdiv, const u64* tab, const u32* dig, u64* res, u32* exc) {
    const u32 i = blockIdx.x * TPB_LOOP + threadIdx.x;
    if (i >= n) return;
    g1_smul_loop_asm(i, n, i / tdiv, n / tdiv, tab, dig, res, exc);
}
__global__ void __launch_bounds__(TPB_EC) k_g1_smul_finish(u32 n, const u64* points, u32 p_stride, u32 p_div, const u64* scalars, u32 s_stride,
                                                            u32 s_div, G1AsmWs ws, u64* out, u32 recompute_flagged, u32 tdiv) {
    const u32 i = blockIdx.x * TPB_EC + threadIdx.x;
    if (i >= n) return;
    if (ws.exc0[i]) { g1_store(out + 12 * (size_t)i, g1_identity()); return; }        // identity in / zero scalar: nothing to compute
    if (ws.exc1[i] && recompute_flagged) {                                   // rare: exact recomputation on the compiled path
        const G1 p = points ? g1_load(points + (size_t)p_stride * (i / p_div)) : g1_generator();
        const Fe s = fe_load(scalars + (size_t)s_stride * (i / s_div));
        g1_store(out + 12 * (size_t)i, g1_scalar_mul_glv5(p, s, ws.jtab, i, n));
        return;
    }
    G1 r = g1_load(ws.res + 12 * (size_t)i);
    r.z = FQ_MUL(r.z, fe_load(ws.zc + 4 * (size_t)(i / tdiv)));
    g1_store(out + 12 * (size_t)i, r);
}
#define G1_ASM_WS_BYTES (16 * G1_ASM29_JT_STRIDE + G1_ASM_TABLE * 64 + G1_ASM_STEPS * 4 + 96 + 32 + 8)
static inline G1AsmWs g1_asm_carve(char* base, size_t n) {
    G1AsmWs w;
    w.jtab = (u64*)base; base += n * 16 * G1_ASM29_JT_STRIDE;     // G1_ASM29_JT_STRIDE = 128 B per entry for the 29-bit table kernel (112 used), 96 used otherwise
    w.tab = (u64*)base; base += n * G1_ASM_TABLE * 64;
    w.res = (u64*)base; base += n * 96;
    w.zc = (u64*)base; base += n * 32;
    w.dig = (u32*)base; base += n * G1_ASM_STEPS * 4;
    w.exc0 = (u32*)base; base += n * 4;
    w.exc1 = (u32*)base;
    return w;
}

// PointShare::add_public (curve/share.rs:57-60): share += rhs iff PARTY0 ; mac += mac_key * rhs
// (mac_key * rhs through the GLV window path: ~2.2 k instead of ~3.8 k Fq multiplications for plain double-and-add)
// NEG: sub_public = add_public(-rhs) (curve/share.rs:63-65)
template <bool NEG>
__global__ void __launch_bounds__(TPB_EC) k_pointshare_add_public(size_t n, int party, Fe key, const u64* shares, const u64* pub, u64* out, u64* table_ws) {
    size_t i = (size_t)blockIdx.x * TPB_EC + threadIdx.x;
    if (i >= n) return;
    G1 rhs = g1_load(pub + 12 * i);
    if (NEG) rhs = g1_neg(rhs);
    G1 sh = g1_load(shares + 24 * i), mac = g1_load(shares + 24 * i + 12);
    if (party == 0) sh = g1_add(sh, rhs);
    mac = g1_add(mac, g1_scalar_mul_glv5(rhs, key, table_ws, i, n));
    g1_store(out + 24 * i, sh);
    g1_store(out + 24 * i + 12, mac);
}
__global__ void __launch_bounds__(64) k_store_scalar(Fe v, u64* out) {          // a host scalar (kernel argument) -> device memory
    if (blockIdx.x | threadIdx.x) return;
    fe_store(out, v);
}
// the same two operations with mac_key * point already computed by the scalar-mul pipeline (kp): additions only
template <bool NEG>
__global__ void __launch_bounds__(TPB_EC) k_pointshare_add_public_kp(size_t n, int party, const u64* shares, const u64* pub, const u64* kp, u64* out) {
    size_t i = (size_t)blockIdx.x * TPB_EC + threadIdx.x;
    if (i >= n) return;
    G1 rhs = g1_load(pub + 12 * i), krhs = g1_load(kp + 12 * i);
    if (NEG) { rhs = g1_neg(rhs); krhs = g1_neg(krhs); }              // key * (-rhs) = -(key * rhs)
    G1 sh = g1_load(shares + 24 * i), mac = g1_load(shares + 24 * i + 12);
    if (party == 0) sh = g1_add(sh, rhs);
    mac = g1_add(mac, krhs);
    g1_store(out + 24 * i, sh);
    g1_store(out + 24 * i + 12, mac);
}
__global__ void __launch_bounds__(TPB_EC) k_point_mac_check_kp(size_t n, const u64* kv, const u64* shares, u64* out) {
    size_t i = (size_t)blockIdx.x * TPB_EC + threadIdx.x;
    if (i >= n) return;
    g1_store(out + 12 * i, g1_add(g1_load(kv + 12 * i), g1_neg(g1_load(shares + 24 * i + 12))));
}
// value * mac_key - share.mac()  (authenticated_curve.rs:215-220)
__global__ void __launch_bounds__(TPB_EC) k_point_mac_check(size_t n, Fe key, const u64* opened, const u64* shares, u64* out, u64* table_ws) {
    size_t i = (size_t)blockIdx.x * TPB_EC + threadIdx.x;
    if (i >= n) return;
    G1 v = g1_load(opened + 12 * i), mac = g1_load(shares + 24 * i + 12);
    g1_store(out + 12 * i, g1_add(g1_scalar_mul_glv5(v, key, table_ws, i, n), g1_neg(mac)));
}
// my + peer == identity  (authenticated_curve.rs:127-131), per element
__global__ void __launch_bounds__(TPB_EC) k_point_mac_verify(size_t n, const u64* mine, const u64* peer, unsigned char* ok) {
    size_t i = (size_t)blockIdx.x * TPB_EC + threadIdx.x;
    if (i >= n) return;
    G1 s = g1_add(g1_load(mine + 12 * i), g1_load(peer + 12 * i));
    ok[i] = FQ_ISZERO(s.z) ? 1 : 0;
}
__global__ void __launch_bounds__(TPB_EC) k_pointshare_extract(size_t n, const u64* shares, u64* out) {
    size_t i = (size_t)blockIdx.x * TPB_EC + threadIdx.x;
    if (i >= n) return;
    g1_store(out + 12 * i, g1_load(shares + 24 * i));
}
__global__ void __launch_bounds__(TPB_EC) k_g1_to_affine(size_t n, const u64* pts, const u64* zinv, u64* out_xy, unsigned char* out_inf) {
    size_t i = (size_t)blockIdx.x * TPB_EC + threadIdx.x;
    if (i >= n) return;
    Fe x, y; bool inf;
    g1_to_affine_zi(g1_load(pts + 12 * i), fe_load(zinv + 4 * i), x, y, inf);
    fe_store(out_xy + 8 * i, x);
    fe_store(out_xy + 8 * i + 4, y);
    out_inf[i] = inf ? 1 : 0;
}
// CurvePoint::to_bytes (curve.rs:103-108) = ark-serialize compressed SW encoding: x little-endian,
// bit 7 of the last byte set iff y > -y (as integers), bit 6 set (and x = 0) for the identity.
__global__ void __launch_bounds__(TPB_EC) k_g1_to_bytes(size_t n, const u64* pts, const u64* zinv, unsigned char* out) {
    size_t i = (size_t)blockIdx.x * TPB_EC + threadIdx.x;
    if (i >= n) return;
    Fe x, y; bool inf;
    g1_to_affine_zi(g1_load(pts + 12 * i), fe_load(zinv + 4 * i), x, y, inf);
    Fe xc = fe_to_canonical<FQ>(x), yc = fe_to_canonical<FQ>(y), nyc = fe_to_canonical<FQ>(FQ_NEG(y));
    // y > -y  <=>  (-y) - y borrows
    u32 br = 0, bo;
#pragma unroll
    for (int k = 0; k < 8; ++k) { (void)__builtin_subc(nyc.v[k], yc.v[k], br, &bo); br = bo; }
    u32 top = xc.v[7];
    if (inf) top |= 0x40000000u;
    else if (br) top |= 0x80000000u;
    uint4* q = reinterpret_cast<uint4*>(out + 32 * i);
    q[0] = make_uint4(xc.v[0], xc.v[1], xc.v[2], xc.v[3]);
    q[1] = make_uint4(xc.v[4], xc.v[5], xc.v[6], top);
}

// CurvePoint::from_bytes (curve.rs:110-114) = deserialize_compressed with validation, the inverse of k_g1_to_bytes:
// x must be canonical (< q) and at most one flag bit set; bit 6 -> identity; otherwise y = (x^3 + 3)^((q+1)/4) (q = 3 mod 4)
// must square back to x^3 + 3, and bit 7 selects the larger root.  ok[i] = 0 (and the identity) for a non-encoding.
__global__ void __launch_bounds__(TPB_EC) k_g1_from_bytes(size_t n, const unsigned char* in, u64* out, unsigned char* ok) {
    size_t i = (size_t)blockIdx.x * TPB_EC + threadIdx.x;
    if (i >= n) return;
    using P = FieldParams<FQ>;
    const uint4* q = reinterpret_cast<const uint4*>(in + 32 * i);
    const uint4 a = q[0], b = q[1];
    Fe xc;
    xc.v[0] = a.x; xc.v[1] = a.y; xc.v[2] = a.z; xc.v[3] = a.w; xc.v[4] = b.x; xc.v[5] = b.y; xc.v[6] = b.z; xc.v[7] = b.w & 0x3fffffffu;
    const bool neg = (b.w >> 31) & 1u, inf = (b.w >> 30) & 1u;
    u32 br = 0, bo;
#pragma unroll
    for (int k = 0; k < 8; ++k) { (void)__builtin_subc(xc.v[k], P::P(k), br, &bo); br = bo; }
    bool valid = br && !(neg && inf);                 // x - q borrows <=> x < q
    G1 r = g1_identity();
    if (valid && !inf) {
        const Fe x = fe_from_canonical<FQ>(xc);
        const Fe three = FQ_ADD(FQ_DBL(fe_one<FQ>()), fe_one<FQ>());
        const Fe rhs = FQ_ADD(FQ_MUL(FQ_SQR(x), x), three);
        u32 e[8], cy = 1;                             // e = (q + 1) / 4
#pragma unroll
        for (int k = 0; k < 8; ++k) { const u64 t = (u64)P::P(k) + cy; e[k] = (u32)t; cy = (u32)(t >> 32); }
#pragma unroll
        for (int k = 0; k < 8; ++k) e[k] = (e[k] >> 2) | (k < 7 ? e[k + 1] << 30 : 0u);
        Fe y = fe_one<FQ>();
        for (int limb = 7; limb >= 0; --limb) {
            u32 w = 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) w = (limb == k) ? e[k] : w;
            for (int bit = 31; bit >= 0; --bit) {
                y = FQ_SQR(y);
                if ((w >> bit) & 1u) y = FQ_MUL(y, rhs);
            }
        }
        valid = FQ_EQ(FQ_SQR(y), rhs);
        const Fe ny = FQ_NEG(y);
        const Fe yc = fe_to_canonical<FQ>(y), nyc = fe_to_canonical<FQ>(ny);
        u32 b2 = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) { (void)__builtin_subc(nyc.v[k], yc.v[k], b2, &bo); b2 = bo; }   // borrows <=> y > -y
        const bool y_is_larger = b2 != 0;
        if (valid) { r.x = x; r.y = (y_is_larger == neg) ? y : ny; r.z = fe_one<FQ>(); }
    }
    g1_store(out + 12 * i, r);
    ok[i] = valid ? 1 : 0;
}

// ---------------------------------------------------------------------------------------------
// Point sums: the reduction gate of AuthenticatedPointResult::msm (authenticated_curve.rs:796-805) and
// PointShare's Sum (curve/share.rs:85-92).  Two launches: every thread folds a strided slice of the n points, then one
// workgroup tree-reduces the partial sums through LDS (256 x 96 B).  `lanes` independent sums are computed at once
// (2 for PointShares: share column and MAC column), selected by blockIdx.y.
// ---------------------------------------------------------------------------------------------
#define SUM_TPB 256
__global__ void __launch_bounds__(SUM_TPB) k_g1_partial_sum(size_t n, const u64* pts, u32 stride, u32 lane_off, u64* partial, u32 nthreads) {
    const u32 lane = blockIdx.y;
    const u32 t = blockIdx.x * SUM_TPB + threadIdx.x;
    if (t >= nthreads) return;
    G1 acc = g1_identity();
    for (size_t i = t; i < n; i += nthreads) acc = g1_add(acc, g1_load(pts + (size_t)stride * i + (size_t)lane_off * lane));
    g1_store(partial + ((size_t)lane * nthreads + t) * 12, acc);
}
__global__ void __launch_bounds__(SUM_TPB) k_g1_final_sum(const u64* partial, u32 nthreads, u64* out) {
    __shared__ u64 sm[SUM_TPB * 12];
    const u32 lane = blockIdx.y, t = threadIdx.x;
    G1 acc = g1_identity();
    for (u32 i = t; i < nthreads; i += SUM_TPB) acc = g1_add(acc, g1_load(partial + ((size_t)lane * nthreads + i) * 12));
    g1_store(sm + 12 * t, acc);
    __syncthreads();
    for (u32 s = SUM_TPB / 2; s > 0; s >>= 1) {
        if (t < s) g1_store(sm + 12 * t, g1_add(g1_load(sm + 12 * t), g1_load(sm + 12 * (t + s))));
        __syncthreads();
    }
    if (t == 0) g1_store(out + 12 * lane, g1_load(sm));
}

// ---------------------------------------------------------------------------------------------
// K9: per-element hash commitments to points -- HashCommitmentResult::commit on each MAC-check point
// (authenticated_curve.rs:227 -> commitment.rs:58-89 with one value):
//     out_i = from_be_bytes_mod_order( SHA3-256( to_bytes(P_i) || to_bytes_be(blinder_i) ) )
// Unlike the scalar batch commitment (one sequential sponge over all values), these are n independent 64-byte
// messages = one Keccak-f[1600] each, so the whole thing runs on the GPU, one thread per commitment.
// ---------------------------------------------------------------------------------------------
#include "keccak_device.inc"
__device__ __forceinline__ u64 limb64(const Fe& f, int i) { return (u64)f.v[2 * i] | ((u64)f.v[2 * i + 1] << 32); }

__global__ void __launch_bounds__(TPB_EC) k_commit_points(size_t n, const u64* pts, const u64* zinv, const u64* blinders, u64* out) {
    size_t i = (size_t)blockIdx.x * TPB_EC + threadIdx.x;
    if (i >= n) return;
    // to_bytes(P): compressed encoding, as four little-endian u64 lanes (curve.rs:103-108)
    Fe x, y; bool inf;
    g1_to_affine_zi(g1_load(pts + 12 * i), fe_load(zinv + 4 * i), x, y, inf);
    Fe xc = fe_to_canonical<FQ>(x), yc = fe_to_canonical<FQ>(y), nyc = fe_to_canonical<FQ>(FQ_NEG(y));
    u32 br = 0, bo;
#pragma unroll
    for (int k = 0; k < 8; ++k) { (void)__builtin_subc(nyc.v[k], yc.v[k], br, &bo); br = bo; }
    if (inf) xc.v[7] |= 0x40000000u; else if (br) xc.v[7] |= 0x80000000u;
    // to_bytes_be(blinder): 32 big-endian bytes = limbs reversed and byte-swapped (scalar.rs:118-127)
    Fe bc = fe_to_canonical<FR>(fe_load(blinders + 4 * i));
    u64 a[25];
#pragma unroll
    for (int k = 0; k < 25; ++k) a[k] = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) { a[k] = limb64(xc, k); a[4 + k] = __builtin_bswap64(limb64(bc, 3 - k)); }
    a[8] ^= 0x06ULL;                    // SHA3 domain separation + first pad bit at byte 64
    a[16] ^= 0x8000000000000000ULL;     // last pad bit at byte 135 (rate = 136)
    keccak_f1600_dev(a);
    // digest bytes (lanes 0..3, little-endian) read as ONE big-endian integer, reduced mod r (scalar.rs:109-112)
    Fe v;
#pragma unroll
    for (int k = 0; k < 4; ++k) { u64 w = __builtin_bswap64(a[3 - k]); v.v[2 * k] = (u32)w; v.v[2 * k + 1] = (u32)(w >> 32); }
    fe_store(out + 4 * i, fe_from_canonical<FR>(fe_reduce_once_loop<FR>(v)));
}

// ---------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------
#define ENTER_EC(ctx)                                                                           \
    if (!(ctx)) return ARKMPC_ERR_BAD_ARG;                                                      \
    CtxGuard guard__(ctx);                                                                      \
    if (guard__.rc) return guard__.rc;                                                          \
    if ((ctx)->field_id != ARKMPC_BN254_FR) { ark_set_err((ctx), "point ops need a BN254_FR context"); return ARKMPC_ERR_UNSUPPORTED; }

static inline bool party_ok(int p) { return p == 0 || p == 1; }
static const size_t EC_CHUNK = (size_t)1 << 20;     // scalar-muls per launch: bounds the window-table workspace at 1.4 GiB

// generator table, one per device, built on first use (a few milliseconds) and kept for the life of the process
static std::mutex g_gen_mu;
static u64* g_gen_table[16] = {nullptr};      // [0, N): entries in the engine's Montgomery form; [N, 2N): the same entries with radix 2^261 (29-bit-limb chain)
static const size_t GEN_TABLE_WORDS = (size_t)GEN_WINDOWS * GEN_ENTRIES * 8;
static int gen_table(arkmpc_ctx* ctx, const u64** out) {
    const int dev = ctx->device;
    if (dev < 0 || dev >= 16) return ark_bad(ctx, "device index");
    std::lock_guard<std::mutex> lk(g_gen_mu);
    if (!g_gen_table[dev]) {
        u64 *bases = nullptr, *table = nullptr;
        ARK_HIP(ctx, hipMalloc((void**)&bases, GEN_WINDOWS * 96));
        ARK_HIP(ctx, hipMalloc((void**)&table, 2 * GEN_TABLE_WORDS * 8));                       // 2 x 2.75 MiB
        hipLaunchKernelGGL(k_gen_table_bases, dim3(1), dim3(64), 0, ctx->stream, bases);
        hipLaunchKernelGGL(k_gen_table_fill, dim3(blocks_for(GEN_WINDOWS * GEN_ENTRIES, TPB_EC)), dim3(TPB_EC), 0, ctx->stream, bases, table, table + GEN_TABLE_WORDS);
        ARK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        ARK_HIP(ctx, hipFree(bases));
        g_gen_table[dev] = table;
    }
    *out = g_gen_table[dev];
    return ARKMPC_OK;
}

// batched z^-1 of n points into scratch (two 32-byte columns: running products, inverses)
static void launch_zinv(arkmpc_ctx* ctx, size_t n, const u64* pts, u64* pre, u64* zinv) {
    size_t k = n >> 15;
    const u32 K = (u32)(k < 8 ? 8 : (k > 64 ? 64 : k));
    const size_t threads = (n + K - 1) / K;
    hipLaunchKernelGGL(k_g1_zinv, dim3(blocks_for(threads, TPB_EC)), dim3(TPB_EC), 0, ctx->stream, n, K, threads, pts, pre, zinv);
}

// The hand-scheduled scalar-mul pipeline over m outputs (out_j = points[j / p_div] * scalars[j / s_div]; strides in u64 units,
// s_stride = 0 broadcasts one scalar), in chunks of G1_ASM_CHUNK; `ws` holds G1_ASM_WS_BYTES per scalar-mul of one chunk.
static const size_t G1_ASM_CHUNK = (size_t)1 << 19;            // 1.6 GB of tables / records per launch
static void g1_smul_launch(arkmpc_ctx* ctx, size_t m, const u64* points, u32 p_stride, u32 p_div, const u64* scalars, u32 s_stride, u32 s_div, u64* out,
                           char* wsbase) {
    const size_t achunk = m < G1_ASM_CHUNK ? m : G1_ASM_CHUNK;
    static const bool asm_prep = !(getenv("ARKMPC_EC_ASM_PREP") && getenv("ARKMPC_EC_ASM_PREP")[0] == '0');
    static const bool limbs29 = g1_limbs29();
    for (size_t lo = 0; lo < m; lo += achunk) {                // chunk boundaries are even: the point / scalar divisors (1 or 2) stay aligned
        const size_t cnt = (m - lo < achunk) ? (m - lo) : achunk;
        const u64* pp = points ? points + (size_t)p_stride * (lo / p_div) : (const u64*)nullptr;
        const u64* sp = scalars + (size_t)s_stride * (lo / s_div);
        const G1AsmWs ws = g1_asm_carve(wsbase, cnt);
        u32 tdiv = 1;
        if (pp && asm_prep) {
            hipLaunchKernelGGL(k_g1_smul_digits, dim3(blocks_for(cnt, TPB_LOOP)), dim3(TPB_LOOP), 0, ctx->stream, (u32)cnt, pp, p_stride, p_div, sp, s_stride, s_div,
                               ws.dig, ws.exc0);
            // p_div lanes multiply the same point: one table column serves them all (the table is a sixth of a scalar-mul's work)
            if (p_div > 1 && cnt % p_div == 0) tdiv = p_div;
            const u32 ncol = (u32)(cnt / tdiv);
            if (limbs29) hipLaunchKernelGGL(k_g1_smul_table29, dim3(blocks_for(ncol, TPB_LOOP)), dim3(TPB_LOOP), 0, ctx->stream, ncol, pp, p_stride, p_div / tdiv, ws.jtab, ws.tab, ws.zc);
            else hipLaunchKernelGGL(k_g1_smul_table, dim3(blocks_for(ncol, TPB_LOOP)), dim3(TPB_LOOP), 0, ctx->stream, ncol, pp, p_stride, p_div / tdiv, ws.jtab, ws.tab, ws.zc);
        } else {
            hipLaunchKernelGGL(k_g1_smul_prep, dim3(blocks_for(cnt, TPB_EC)), dim3(TPB_EC), 0, ctx->stream, (u32)cnt, pp, p_stride, p_div, sp, s_stride, s_div, ws);
        }
        if (limbs29 && pp && asm_prep) hipLaunchKernelGGL(k_g1_smul_loop29, dim3(blocks_for(cnt, TPB_LOOP)), dim3(TPB_LOOP), 0, ctx->stream, (u32)cnt, tdiv, ws.tab, ws.dig, ws.res, ws.exc1);
        else hipLaunchKernelGGL(k_g1_smul_loop, dim3(blocks_for(cnt, TPB_LOOP)), dim3(TPB_LOOP), 0, ctx->stream, (u32)cnt, tdiv, ws.tab, ws.dig, ws.res, ws.exc1);
        // test hook: ARKMPC_EC_ASM_NOFIX=1 leaves flagged lanes as the loop produced them (garbage), which is how the tests prove that the
        // crafted inputs really reach the exceptional path
        static const u32 recompute = (getenv("ARKMPC_EC_ASM_NOFIX") && getenv("ARKMPC_EC_ASM_NOFIX")[0] == '1') ? 0u : 1u;
        hipLaunchKernelGGL(k_g1_smul_finish, dim3(blocks_for(cnt, TPB_EC)), dim3(TPB_EC), 0, ctx->stream, (u32)cnt, pp, p_stride, p_div, sp, s_stride, s_div, ws,
                           out + 12 * lo, recompute, tdiv);
    }
}
static inline size_t g1_smul_ws_bytes(size_t m) { return (m < G1_ASM_CHUNK ? m : G1_ASM_CHUNK) * G1_ASM_WS_BYTES + 256; }
static bool g1_asm_enabled() {
    static const bool on = !(getenv("ARKMPC_EC_ASM") && getenv("ARKMPC_EC_ASM")[0] == '0');
    return on;
}

extern "C" {

static int g1_addsub(arkmpc_ctx* ctx, bool sub, size_t n, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t mult) {
    ENTER_EC(ctx);
    Stage st(ctx);
    const size_t m = n * mult;  // mult = 2 treats n PointShares as 2n points
    int ia = st.declare_in(a, m * 96), ib = st.declare_in(b, m * 96), io = st.declare_out(out, m * 96);
    if (st.commit()) return st.rc;
    if (m) {
        dim3 g(blocks_for(m, TPB_EC)), t(TPB_EC);
        if (sub) hipLaunchKernelGGL((k_g1_add<true>), g, t, 0, ctx->stream, m, st.in<u64>(ia), st.in<u64>(ib), st.out<u64>(io));
        else hipLaunchKernelGGL((k_g1_add<false>), g, t, 0, ctx->stream, m, st.in<u64>(ia), st.in<u64>(ib), st.out<u64>(io));
    }
    return st.finish();
}
int arkmpc_g1_add(arkmpc_ctx* ctx, size_t n, const uint64_t* a, const uint64_t* b, uint64_t* out) { return g1_addsub(ctx, false, n, a, b, out, 1); }
int arkmpc_g1_sub(arkmpc_ctx* ctx, size_t n, const uint64_t* a, const uint64_t* b, uint64_t* out) { return g1_addsub(ctx, true, n, a, b, out, 1); }
int arkmpc_pointshare_add(arkmpc_ctx* ctx, size_t n, const uint64_t* a, const uint64_t* b, uint64_t* out) { return g1_addsub(ctx, false, n, a, b, out, 2); }
int arkmpc_pointshare_sub(arkmpc_ctx* ctx, size_t n, const uint64_t* a, const uint64_t* b, uint64_t* out) { return g1_addsub(ctx, true, n, a, b, out, 2); }

static int g1_neg_impl(arkmpc_ctx* ctx, size_t m, const uint64_t* a, uint64_t* out) {
    ENTER_EC(ctx);
    Stage st(ctx);
    int ia = st.declare_in(a, m * 96), io = st.declare_out(out, m * 96);
    if (st.commit()) return st.rc;
    if (m) hipLaunchKernelGGL(k_g1_neg, dim3(blocks_for(m, TPB_EC)), dim3(TPB_EC), 0, ctx->stream, m, st.in<u64>(ia), st.out<u64>(io));
    return st.finish();
}
int arkmpc_g1_neg(arkmpc_ctx* ctx, size_t n, const uint64_t* a, uint64_t* out) { return g1_neg_impl(ctx, n, a, out); }
int arkmpc_pointshare_neg(arkmpc_ctx* ctx, size_t n, const uint64_t* a, uint64_t* out) { return g1_neg_impl(ctx, 2 * n, a, out); }

// generic scalar-mul launcher: m output points; point index = i / p_div, scalar index = i / s_div.
// Batches are processed in chunks of 2^20 scalar-muls so the window-table workspace stays at 1.4 GiB.
static int smul_impl(arkmpc_ctx* ctx, size_t m, const uint64_t* points, size_t n_points, u32 p_stride, u32 p_div,
                     const uint64_t* scalars, size_t scalar_bytes, u32 s_stride, u32 s_div, uint64_t* out) {
    ENTER_EC(ctx);
    Stage st(ctx);
    int ip = points ? st.declare_in(points, n_points * 96) : -1;
    int is = st.declare_in(scalars, scalar_bytes), io = st.declare_out(out, m * 96);
    const size_t CH = (size_t)1 << 20;
    const size_t chunk = m < CH ? m : CH;
    static const bool fixed_base = !(getenv("ARKMPC_NO_FIXED_BASE") && getenv("ARKMPC_NO_FIXED_BASE")[0] == '1');
    // variable-base path: the hand-scheduled window loop (prep / loop / finish kernels) unless ARKMPC_EC_ASM=0
    const bool asm_loop = g1_asm_enabled();
    int iw = -1, igv = -1, igl = -1, ige = -1;
    if (points || !fixed_base) iw = asm_loop ? st.declare_scratch(g1_smul_ws_bytes(m)) : st.declare_scratch(chunk * 16 * 96);
    else { igv = st.declare_scratch(chunk * GEN_WINDOWS * 4 + 64); igl = st.declare_scratch(chunk * 4 + 64); ige = st.declare_scratch(chunk * 4 + 64); }
    if (st.commit()) return st.rc;
    if (m && !points && fixed_base) {          // multiplication by the generator: tabulated multiples, no doublings
        const u64* table = nullptr;
        int rc = gen_table(ctx, &table);
        if (rc) return rc;
        static const bool fix = !(getenv("ARKMPC_EC_ASM_NOFIX") && getenv("ARKMPC_EC_ASM_NOFIX")[0] == '1');
        for (size_t lo = 0; lo < m; lo += chunk) {
            const size_t cnt = (m - lo < chunk) ? (m - lo) : chunk;
            const u64* sp = st.in<u64>(is) + (size_t)s_stride * (lo / s_div);
            u32 *vals = st.scratch<u32>(igv), *lens = st.scratch<u32>(igl), *exc = st.scratch<u32>(ige);
            u64* o = st.out<u64>(io) + 12 * lo;
            hipLaunchKernelGGL(k_gen_digits, dim3(blocks_for(cnt, 256)), dim3(256), 0, ctx->stream, cnt, sp, s_stride, s_div, vals, lens);
            if (asm_loop) {
                if (g1_limbs29()) hipLaunchKernelGGL(k_gen_mul_chain_asm29, dim3(blocks_for(cnt, TPB_LOOP)), dim3(TPB_LOOP), 0, ctx->stream, (u32)cnt, vals, lens, table + GEN_TABLE_WORDS, o, exc);
                else hipLaunchKernelGGL(k_gen_mul_chain_asm, dim3(blocks_for(cnt, TPB_LOOP)), dim3(TPB_LOOP), 0, ctx->stream, (u32)cnt, vals, lens, table, o, exc);
                if (fix) hipLaunchKernelGGL(k_gen_mul_chain, dim3(blocks_for(cnt, TPB_EC)), dim3(TPB_EC), 0, ctx->stream, cnt, vals, lens, table, o, (const u32*)exc);
            } else {
                hipLaunchKernelGGL(k_gen_mul_chain, dim3(blocks_for(cnt, TPB_EC)), dim3(TPB_EC), 0, ctx->stream, cnt, vals, lens, table, o, (const u32*)nullptr);
            }
        }
        return st.finish();
    }
    if (m && asm_loop) {
        g1_smul_launch(ctx, m, points ? st.in<u64>(ip) : (const u64*)nullptr, p_stride, p_div, st.in<u64>(is), s_stride, s_div, st.out<u64>(io), st.scratch<char>(iw));
        return st.finish();
    }
    if (m) {
        // chunk boundaries must respect the point / scalar divisors (1 or 2)
        for (size_t lo = 0; lo < m; lo += chunk) {
            const size_t cnt = (m - lo < chunk) ? (m - lo) : chunk;
            const u64* pp = points ? st.in<u64>(ip) + (size_t)p_stride * (lo / p_div) : (const u64*)nullptr;
            const u64* sp = st.in<u64>(is) + (size_t)s_stride * (lo / s_div);
            // 2 = GLV with signed 5-bit windows (default), 1 = GLV with unsigned 4-bit windows (ARKMPC_GLV_W=4), 0 = no endomorphism
            static const int glv = (getenv("ARKMPC_NO_GLV") && getenv("ARKMPC_NO_GLV")[0] == '1') ? 0 : ((getenv("ARKMPC_GLV_W") && getenv("ARKMPC_GLV_W")[0] == '4') ? 1 : 2);
            const dim3 g(blocks_for(cnt, TPB_EC)), t(TPB_EC);
            u64* op = st.out<u64>(io) + 12 * lo;
            if (glv == 2) hipLaunchKernelGGL(k_g1_scalar_mul<2>, g, t, 0, ctx->stream, cnt, pp, p_stride, p_div, sp, s_stride, s_div, op, st.scratch<u64>(iw));
            else if (glv == 1) hipLaunchKernelGGL(k_g1_scalar_mul<1>, g, t, 0, ctx->stream, cnt, pp, p_stride, p_div, sp, s_stride, s_div, op, st.scratch<u64>(iw));
            else hipLaunchKernelGGL(k_g1_scalar_mul<0>, g, t, 0, ctx->stream, cnt, pp, p_stride, p_div, sp, s_stride, s_div, op, st.scratch<u64>(iw));
        }
    }
    return st.finish();
}
int arkmpc_g1_scalar_mul(arkmpc_ctx* ctx, size_t n, const uint64_t* points, const uint64_t* scalars, uint64_t* out) {
    if (n && !points) return ctx ? ark_bad(ctx, "null points") : ARKMPC_ERR_BAD_ARG;
    return smul_impl(ctx, n, points, n, 12, 1, scalars, n * 32, 4, 1, out);
}
int arkmpc_g1_generator_mul(arkmpc_ctx* ctx, size_t n, const uint64_t* scalars, uint64_t* out) {
    return smul_impl(ctx, n, nullptr, 0, 0, 1, scalars, n * 32, 4, 1, out);
}
// PointShare * Scalar: output point j (0..2n) = shares_as_points[j] * scalars[j / 2]
int arkmpc_pointshare_mul_public(arkmpc_ctx* ctx, size_t n, const uint64_t* shares, const uint64_t* scalars, uint64_t* out) {
    if (n && !shares) return ctx ? ark_bad(ctx, "null shares") : ARKMPC_ERR_BAD_ARG;
    return smul_impl(ctx, 2 * n, shares, 2 * n, 12, 1, scalars, n * 32, 4, 2, out);
}
// ScalarShare * generator: output point j = G * scalar_shares_as_scalars[j]
int arkmpc_scalarshare_mul_generator(arkmpc_ctx* ctx, size_t n, const uint64_t* scalar_shares, uint64_t* out) {
    return smul_impl(ctx, 2 * n, nullptr, 0, 0, 1, scalar_shares, n * 64, 4, 1, out);
}
// ScalarShare * CurvePoint: output point j = points[j / 2] * scalar_shares_as_scalars[j]
int arkmpc_scalarshare_mul_point(arkmpc_ctx* ctx, size_t n, const uint64_t* scalar_shares, const uint64_t* points, uint64_t* out) {
    if (n && !points) return ctx ? ark_bad(ctx, "null points") : ARKMPC_ERR_BAD_ARG;
    return smul_impl(ctx, 2 * n, points, n, 12, 2, scalar_shares, n * 64, 4, 1, out);
}

static int pointshare_addsub_public(arkmpc_ctx* ctx, bool sub, size_t n, int party_id, const uint64_t mac_key[4], const uint64_t* shares,
                                    const uint64_t* pub_points, uint64_t* out) {
    ENTER_EC(ctx);
    if (!party_ok(party_id)) return ark_bad(ctx, "party_id must be 0 or 1");
    if (!mac_key) return ark_bad(ctx, "null mac_key");
    Stage st(ctx);
    int is = st.declare_in(shares, n * 192), ip = st.declare_in(pub_points, n * 96), io = st.declare_out(out, n * 192);
    const bool asm_loop = g1_asm_enabled();
    const size_t chunk = n < EC_CHUNK ? n : EC_CHUNK;
    int iw = asm_loop ? st.declare_scratch(g1_smul_ws_bytes(n)) : st.declare_scratch(chunk * 16 * 96);
    int ik = asm_loop ? st.declare_scratch(n * 96 + 64) : -1;          // mac_key (32 B) + n points mac_key * rhs
    if (st.commit()) return st.rc;
    if (n && asm_loop) {                                               // mac_key * rhs through the scalar-mul pipeline (one broadcast scalar)
        u64* dkey = st.scratch<u64>(ik);
        u64* kp = dkey + 8;
        hipLaunchKernelGGL(k_store_scalar, dim3(1), dim3(64), 0, ctx->stream, fe_from_host(mac_key), dkey);
        g1_smul_launch(ctx, n, st.in<u64>(ip), 12, 1, dkey, 0, 1, kp, st.scratch<char>(iw));
        const dim3 g(blocks_for(n, TPB_EC)), t(TPB_EC);
        if (sub) hipLaunchKernelGGL(k_pointshare_add_public_kp<true>, g, t, 0, ctx->stream, n, party_id, st.in<u64>(is), st.in<u64>(ip), kp, st.out<u64>(io));
        else hipLaunchKernelGGL(k_pointshare_add_public_kp<false>, g, t, 0, ctx->stream, n, party_id, st.in<u64>(is), st.in<u64>(ip), kp, st.out<u64>(io));
        return st.finish();
    }
    for (size_t lo = 0; lo < n; lo += chunk) {
        const size_t cnt = (n - lo < chunk) ? (n - lo) : chunk;
        const dim3 g(blocks_for(cnt, TPB_EC)), t(TPB_EC);
        if (sub) hipLaunchKernelGGL(k_pointshare_add_public<true>, g, t, 0, ctx->stream, cnt, party_id, fe_from_host(mac_key),
                                    st.in<u64>(is) + 24 * lo, st.in<u64>(ip) + 12 * lo, st.out<u64>(io) + 24 * lo, st.scratch<u64>(iw));
        else hipLaunchKernelGGL(k_pointshare_add_public<false>, g, t, 0, ctx->stream, cnt, party_id, fe_from_host(mac_key),
                                st.in<u64>(is) + 24 * lo, st.in<u64>(ip) + 12 * lo, st.out<u64>(io) + 24 * lo, st.scratch<u64>(iw));
    }
    return st.finish();
}
int arkmpc_pointshare_add_public(arkmpc_ctx* ctx, size_t n, int party_id, const uint64_t mac_key[4], const uint64_t* shares,
                                 const uint64_t* pub_points, uint64_t* out) {
    return pointshare_addsub_public(ctx, false, n, party_id, mac_key, shares, pub_points, out);
}
int arkmpc_pointshare_sub_public(arkmpc_ctx* ctx, size_t n, int party_id, const uint64_t mac_key[4], const uint64_t* shares,
                                 const uint64_t* pub_points, uint64_t* out) {
    return pointshare_addsub_public(ctx, true, n, party_id, mac_key, shares, pub_points, out);
}
int arkmpc_pointshare_extract(arkmpc_ctx* ctx, size_t n, const uint64_t* shares, uint64_t* out_points) {
    ENTER_EC(ctx);
    Stage st(ctx);
    int is = st.declare_in(shares, n * 192), io = st.declare_out(out_points, n * 96);
    if (st.commit()) return st.rc;
    if (n) hipLaunchKernelGGL(k_pointshare_extract, dim3(blocks_for(n, TPB_EC)), dim3(TPB_EC), 0, ctx->stream, n, st.in<u64>(is), st.out<u64>(io));
    return st.finish();
}
int arkmpc_point_mac_check_shares(arkmpc_ctx* ctx, size_t n, const uint64_t mac_key[4], const uint64_t* opened_points,
                                  const uint64_t* shares, uint64_t* out_chk_points) {
    ENTER_EC(ctx);
    if (!mac_key) return ark_bad(ctx, "null mac_key");
    Stage st(ctx);
    int iv = st.declare_in(opened_points, n * 96), is = st.declare_in(shares, n * 192), io = st.declare_out(out_chk_points, n * 96);
    const bool asm_loop = g1_asm_enabled();
    const size_t chunk = n < EC_CHUNK ? n : EC_CHUNK;
    int iw = asm_loop ? st.declare_scratch(g1_smul_ws_bytes(n)) : st.declare_scratch(chunk * 16 * 96);
    int ik = asm_loop ? st.declare_scratch(n * 96 + 64) : -1;
    if (st.commit()) return st.rc;
    if (n && asm_loop) {                                               // value * mac_key through the scalar-mul pipeline, then - mac
        u64* dkey = st.scratch<u64>(ik);
        u64* kv = dkey + 8;
        hipLaunchKernelGGL(k_store_scalar, dim3(1), dim3(64), 0, ctx->stream, fe_from_host(mac_key), dkey);
        g1_smul_launch(ctx, n, st.in<u64>(iv), 12, 1, dkey, 0, 1, kv, st.scratch<char>(iw));
        hipLaunchKernelGGL(k_point_mac_check_kp, dim3(blocks_for(n, TPB_EC)), dim3(TPB_EC), 0, ctx->stream, n, kv, st.in<u64>(is), st.out<u64>(io));
        return st.finish();
    }
    for (size_t lo = 0; lo < n; lo += chunk) {
        const size_t cnt = (n - lo < chunk) ? (n - lo) : chunk;
        hipLaunchKernelGGL(k_point_mac_check, dim3(blocks_for(cnt, TPB_EC)), dim3(TPB_EC), 0, ctx->stream, cnt, fe_from_host(mac_key),
                           st.in<u64>(iv) + 12 * lo, st.in<u64>(is) + 24 * lo, st.out<u64>(io) + 12 * lo, st.scratch<u64>(iw));
    }
    return st.finish();
}
int arkmpc_point_mac_verify(arkmpc_ctx* ctx, size_t n, const uint64_t* mine, const uint64_t* peer, uint8_t* out_ok) {
    ENTER_EC(ctx);
    Stage st(ctx);
    int im = st.declare_in(mine, n * 96), ip = st.declare_in(peer, n * 96), io = st.declare_out(out_ok, n);
    if (st.commit()) return st.rc;
    if (n) hipLaunchKernelGGL(k_point_mac_verify, dim3(blocks_for(n, TPB_EC)), dim3(TPB_EC), 0, ctx->stream, n, st.in<u64>(im), st.in<u64>(ip),
                              st.out<unsigned char>(io));
    return st.finish();
}
static int g1_sum_impl(arkmpc_ctx* ctx, size_t n, const uint64_t* pts, u32 stride, u32 lanes, uint64_t* out) {
    ENTER_EC(ctx);
    Stage st(ctx);
    int ip = st.declare_in(pts, n * stride * 8), io = st.declare_out(out, (size_t)lanes * 96);
    // three short levels instead of one long one: every thread of a level folds at most ~16 inputs (a fold is a DEPENDENT chain of
    // additions, ~5 us each), so 2^18 points take ~40 dependent additions instead of ~270 (2.98 -> 0.3 ms on BN254)
    size_t t1 = n / 8; t1 = t1 < 256 ? (n ? (n < 256 ? n : 256) : 1) : (t1 > 65536 ? 65536 : t1);
    const u32 n1 = (u32)t1, n2 = n1 > 2048 ? n1 / 16 : 0;
    int iw = st.declare_scratch((size_t)lanes * n1 * 96), iw2 = st.declare_scratch((size_t)lanes * (n2 ? n2 : 1) * 96);
    if (st.commit()) return st.rc;
    hipLaunchKernelGGL(k_g1_partial_sum, dim3(blocks_for(n1, SUM_TPB), lanes), dim3(SUM_TPB), 0, ctx->stream, n, st.in<u64>(ip), stride, 12u,
                       st.scratch<u64>(iw), n1);
    const u64* last = st.scratch<u64>(iw);
    u32 nlast = n1;
    if (n2) {                                               // level 2 reads level 1's per-lane arrays: stride = one point, lane offset = one array
        hipLaunchKernelGGL(k_g1_partial_sum, dim3(blocks_for(n2, SUM_TPB), lanes), dim3(SUM_TPB), 0, ctx->stream, (size_t)n1, last, 12u, n1 * 12u,
                           st.scratch<u64>(iw2), n2);
        last = st.scratch<u64>(iw2); nlast = n2;
    }
    hipLaunchKernelGGL(k_g1_final_sum, dim3(1, lanes), dim3(SUM_TPB), 0, ctx->stream, last, nlast, st.out<u64>(io));
    return st.finish();
}
int arkmpc_g1_sum(arkmpc_ctx* ctx, size_t n, const uint64_t* points, uint64_t* out_point) { return g1_sum_impl(ctx, n, points, 12, 1, out_point); }
int arkmpc_pointshare_sum(arkmpc_ctx* ctx, size_t n, const uint64_t* shares, uint64_t* out_share) { return g1_sum_impl(ctx, n, shares, 24, 2, out_share); }
int arkmpc_commit_points_sha3(arkmpc_ctx* ctx, size_t n, const uint64_t* points, const uint64_t* blinders, uint64_t* out_commitments) {
    ENTER_EC(ctx);
    Stage st(ctx);
    int ip = st.declare_in(points, n * 96), ib = st.declare_in(blinders, n * 32), io = st.declare_out(out_commitments, n * 32);
    int iz = st.declare_scratch(n * 64 + 64);
    if (st.commit()) return st.rc;
    if (n) {
        launch_zinv(ctx, n, st.in<u64>(ip), st.scratch<u64>(iz), st.scratch<u64>(iz) + 4 * n);
        hipLaunchKernelGGL(k_commit_points, dim3(blocks_for(n, TPB_EC)), dim3(TPB_EC), 0, ctx->stream, n, st.in<u64>(ip), st.scratch<u64>(iz) + 4 * n,
                           st.in<u64>(ib), st.out<u64>(io));
    }
    return st.finish();
}
int arkmpc_g1_to_affine(arkmpc_ctx* ctx, size_t n, const uint64_t* points, uint64_t* out_xy, uint8_t* out_inf) {
    ENTER_EC(ctx);
    Stage st(ctx);
    int ip = st.declare_in(points, n * 96), io = st.declare_out(out_xy, n * 64), ii = st.declare_out(out_inf, n);
    int iz = st.declare_scratch(n * 64 + 64);
    if (st.commit()) return st.rc;
    if (n) {
        launch_zinv(ctx, n, st.in<u64>(ip), st.scratch<u64>(iz), st.scratch<u64>(iz) + 4 * n);
        hipLaunchKernelGGL(k_g1_to_affine, dim3(blocks_for(n, TPB_EC)), dim3(TPB_EC), 0, ctx->stream, n, st.in<u64>(ip), st.scratch<u64>(iz) + 4 * n,
                           st.out<u64>(io), st.out<unsigned char>(ii));
    }
    return st.finish();
}
int arkmpc_g1_to_bytes(arkmpc_ctx* ctx, size_t n, const uint64_t* points, uint8_t* out_bytes) {
    ENTER_EC(ctx);
    Stage st(ctx);
    int ip = st.declare_in(points, n * 96), io = st.declare_out(out_bytes, n * 32);
    int iz = st.declare_scratch(n * 64 + 64);
    if (st.commit()) return st.rc;
    if (n) {
        launch_zinv(ctx, n, st.in<u64>(ip), st.scratch<u64>(iz), st.scratch<u64>(iz) + 4 * n);
        hipLaunchKernelGGL(k_g1_to_bytes, dim3(blocks_for(n, TPB_EC)), dim3(TPB_EC), 0, ctx->stream, n, st.in<u64>(ip), st.scratch<u64>(iz) + 4 * n,
                           st.out<unsigned char>(io));
    }
    return st.finish();
}

int arkmpc_g1_from_bytes(arkmpc_ctx* ctx, size_t n, const uint8_t* bytes, uint64_t* out_points, uint8_t* out_ok) {
    ENTER_EC(ctx);
    Stage st(ctx);
    int ib = st.declare_in(bytes, n * 32), io = st.declare_out(out_points, n * 96), ik = st.declare_out(out_ok, n);
    if (st.commit()) return st.rc;
    if (n) hipLaunchKernelGGL(k_g1_from_bytes, dim3(blocks_for(n, TPB_EC)), dim3(TPB_EC), 0, ctx->stream, n, st.in<unsigned char>(ib), st.out<u64>(io),
                              st.out<unsigned char>(ik));
    return st.finish();
}

}  // extern "C"

// variable-base MSM (bucket method): CurvePoint::msm / msm_authenticated
#include "arkmpc_msm.inc"
