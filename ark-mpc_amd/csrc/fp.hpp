// fp.hpp -- 256-bit prime-field arithmetic for gfx950 (CDNA4), device side.
//
// Replaces, for the hot path, what ark-mpc delegates to arkworks `Fp256<MontBackend<_,4>>`
// (reference call sites: online-phase/src/algebra/scalar/scalar.rs:215,235,255,264).
// In-memory form is arkworks': 4 x u64 little-endian limbs, Montgomery form, R = 2^256, always
// fully reduced to [0, p).  In registers an element is 8 x u32 limbs (one VGPR each); the
// multiplier is built on v_mad_u64_u32 (32x32+64 -> 64), measured on MI355X at ~1.8x the issue
// cost of a v_add_u32 (profiles/ubench_r01.log), i.e. not the quarter-rate op it was on GCN.
//
// Everything is plain C++ on purpose: gfx950 needs 2 wait states between a VALU that writes
// VCC/SGPR and a VALU that reads it as carry-in, and hipcc pads those itself for compiler-
// generated carry chains but NOT inside inline asm.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint32_t u32;
typedef uint64_t u64;

template <int FID> struct FieldParams;
#include "field_consts.inc"

enum { F_BN254_FR = 0, F_BLS12_381_FR = 1, F_CURVE25519_FR = 2, F_BN254_FQ = 3, F_CURVE25519_FQ = 4, F_NFIELDS = 5 };

struct Fe {
    u32 v[8];
};

// ---- memory access: one element = 32 B = two 16-B vector accesses --------------------------
__device__ __forceinline__ Fe fe_load(const u64* p) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
    uint4 lo = q[0], hi = q[1];
    Fe r;
    r.v[0] = lo.x; r.v[1] = lo.y; r.v[2] = lo.z; r.v[3] = lo.w;
    r.v[4] = hi.x; r.v[5] = hi.y; r.v[6] = hi.z; r.v[7] = hi.w;
    return r;
}
__device__ __forceinline__ void fe_store(u64* p, const Fe& a) {
    uint4* q = reinterpret_cast<uint4*>(p);
    q[0] = make_uint4(a.v[0], a.v[1], a.v[2], a.v[3]);
    q[1] = make_uint4(a.v[4], a.v[5], a.v[6], a.v[7]);
}

// non-temporal forms for data that is streamed exactly once (keeps it from displacing reusable lines in L2 / MALL)
typedef u32 u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ Fe fe_load_nt(const u64* p) {
    const u32x4_t* q = reinterpret_cast<const u32x4_t*>(p);
    u32x4_t lo = __builtin_nontemporal_load(q), hi = __builtin_nontemporal_load(q + 1);
    Fe r;
    r.v[0] = lo.x; r.v[1] = lo.y; r.v[2] = lo.z; r.v[3] = lo.w;
    r.v[4] = hi.x; r.v[5] = hi.y; r.v[6] = hi.z; r.v[7] = hi.w;
    return r;
}
__device__ __forceinline__ void fe_store_nt(u64* p, const Fe& a) {
    u32x4_t* q = reinterpret_cast<u32x4_t*>(p);
    u32x4_t lo = {a.v[0], a.v[1], a.v[2], a.v[3]}, hi = {a.v[4], a.v[5], a.v[6], a.v[7]};
    __builtin_nontemporal_store(lo, q);
    __builtin_nontemporal_store(hi, q + 1);
}

template <int F> __host__ __device__ __forceinline__ Fe fe_zero() {
    Fe r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.v[i] = 0;
    return r;
}
template <int F> __host__ __device__ __forceinline__ Fe fe_one() {  // Montgomery form of 1
    Fe r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.v[i] = FieldParams<F>::ONE(i);
    return r;
}
__host__ __device__ __forceinline__ bool fe_is_zero(const Fe& a) {
    u32 o = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) o |= a.v[i];
    return o == 0;
}
__host__ __device__ __forceinline__ bool fe_eq(const Fe& a, const Fe& b) {
    u32 o = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) o |= a.v[i] ^ b.v[i];
    return o == 0;
}
__host__ __device__ __forceinline__ Fe fe_select(bool c, const Fe& a, const Fe& b) {  // c ? a : b
    Fe r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.v[i] = c ? a.v[i] : b.v[i];
    return r;
}

// ---- add / sub / neg: canonical in, canonical out (all four moduli are < 2^255) -------------
template <int F> __host__ __device__ __forceinline__ Fe fe_add(const Fe& a, const Fe& b) {
    using P = FieldParams<F>;
    u32 s[8], d[8], c = 0, co, br = 0, bo;
#pragma unroll
    for (int i = 0; i < 8; ++i) { s[i] = __builtin_addc(a.v[i], b.v[i], c, &co); c = co; }
#pragma unroll
    for (int i = 0; i < 8; ++i) { d[i] = __builtin_subc(s[i], P::P(i), br, &bo); br = bo; }
    Fe r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.v[i] = br ? s[i] : d[i];
    return r;
}
template <int F> __host__ __device__ __forceinline__ Fe fe_sub(const Fe& a, const Fe& b) {
    using P = FieldParams<F>;
    u32 d[8], br = 0, bo, c = 0, co;
#pragma unroll
    for (int i = 0; i < 8; ++i) { d[i] = __builtin_subc(a.v[i], b.v[i], br, &bo); br = bo; }
    const u32 mask = 0u - br;
    Fe r;
#pragma unroll
    for (int i = 0; i < 8; ++i) { r.v[i] = __builtin_addc(d[i], P::P(i) & mask, c, &co); c = co; }
    return r;
}
template <int F> __host__ __device__ __forceinline__ Fe fe_neg(const Fe& a) {
    using P = FieldParams<F>;
    u32 d[8], br = 0, bo, nz = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) { d[i] = __builtin_subc(P::P(i), a.v[i], br, &bo); br = bo; nz |= a.v[i]; }
    Fe r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.v[i] = nz ? d[i] : 0u;
    return r;
}
template <int F> __host__ __device__ __forceinline__ Fe fe_dbl(const Fe& a) { return fe_add<F>(a, a); }

// ---- Montgomery multiplication: CIOS over 32-bit limbs, one multiplier row + one reduction row
// per outer step.  Row form: q_j = x_j * y + t_j (8 independent v_mad_u64_u32), then one carry
// chain folds hi(q_{j-1}) into lo(q_j).  Output < 2p before the final conditional subtraction.
template <int F> __host__ __device__ __forceinline__ Fe fe_mul(const Fe& a, const Fe& b) {
    using P = FieldParams<F>;
    u32 t[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) t[i] = 0;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        u64 q[8];
        u32 c = 0, co;
#pragma unroll
        for (int j = 0; j < 8; ++j) q[j] = (u64)a.v[j] * b.v[r] + t[j];
        t[0] = (u32)q[0];
#pragma unroll
        for (int j = 1; j < 8; ++j) { t[j] = __builtin_addc((u32)q[j], (u32)(q[j - 1] >> 32), c, &co); c = co; }
        t[8] = __builtin_addc(t[8], (u32)(q[7] >> 32), c, &co);
        t[9] = co;
        const u32 m = t[0] * P::INV32;
#pragma unroll
        for (int j = 0; j < 8; ++j) q[j] = (u64)m * P::P(j) + t[j];
        c = 0;
#pragma unroll
        for (int j = 1; j < 8; ++j) { t[j - 1] = __builtin_addc((u32)q[j], (u32)(q[j - 1] >> 32), c, &co); c = co; }
        t[7] = __builtin_addc(t[8], (u32)(q[7] >> 32), c, &co);
        t[8] = t[9] + co;
    }
    // t < 2p < 2^256, so t[8] == 0 here; one conditional subtraction canonicalises
    u32 d[8], br = 0, bo;
#pragma unroll
    for (int i = 0; i < 8; ++i) { d[i] = __builtin_subc(t[i], P::P(i), br, &bo); br = bo; }
    Fe o;
#pragma unroll
    for (int i = 0; i < 8; ++i) o.v[i] = br ? t[i] : d[i];
    return o;
}
template <int F> __host__ __device__ __forceinline__ Fe fe_sqr(const Fe& a) { return fe_mul<F>(a, a); }

// Montgomery <-> canonical
template <int F> __host__ __device__ __forceinline__ Fe fe_to_canonical(const Fe& a) {
    Fe one;
#pragma unroll
    for (int i = 0; i < 8; ++i) one.v[i] = (i == 0) ? 1u : 0u;
    return fe_mul<F>(a, one);
}
template <int F> __host__ __device__ __forceinline__ Fe fe_from_canonical(const Fe& a) {  // a must be < p
    Fe r2;
#pragma unroll
    for (int i = 0; i < 8; ++i) r2.v[i] = FieldParams<F>::RSQ(i);
    return fe_mul<F>(a, r2);
}
// bring an arbitrary 256-bit value below p by repeated conditional subtraction (p >= 2^252 -> <= 15 rounds)
template <int F> __host__ __device__ __forceinline__ Fe fe_reduce_once_loop(Fe a) {
    using P = FieldParams<F>;
    for (int it = 0; it < 16; ++it) {
        u32 d[8], br = 0, bo;
#pragma unroll
        for (int i = 0; i < 8; ++i) { d[i] = __builtin_subc(a.v[i], P::P(i), br, &bo); br = bo; }
        if (br) break;
#pragma unroll
        for (int i = 0; i < 8; ++i) a.v[i] = d[i];
    }
    return a;
}

// a^(p-2) (Fermat); 0 -> 0.  The exponent limbs are derived from P with the borrow propagated.
template <int F> __device__ __forceinline__ Fe fe_inv_fermat(const Fe& a) {
    using P = FieldParams<F>;
    Fe acc = fe_one<F>();
    for (int limb = 7; limb >= 0; --limb) {
        // limb of p - 2 with the borrow propagated (BLS12-381 Fr ends in ...00000001)
        u32 w = P::P(limb);
        bool borrow = true;   // subtracting 2 from limb 0
        u32 sub = 2u;
        for (int l = 0; l <= limb; ++l) {
            const u32 pl = P::P(l);
            const u32 s = (l == 0) ? sub : (borrow ? 1u : 0u);
            if (l == limb) w = pl - s;
            borrow = pl < s;
        }
        for (int bit = 31; bit >= 0; --bit) {
            acc = fe_sqr<F>(acc);
            if ((w >> bit) & 1u) acc = fe_mul<F>(acc, a);
        }
    }
    return acc;
}
