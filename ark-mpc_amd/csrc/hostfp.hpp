// hostfp.hpp -- host-side Montgomery arithmetic on 4 x u64 limbs (unsigned __int128 products) for the few sequential tails that run on a CPU
// core: the Horner fold over the MSM's window sums (~270 dependent doublings + the per-bit sums).  fp.hpp's Fe arithmetic is host-capable too, but
// it is the GPU's 8 x u32 formulation: ~0.13 us per multiplication on the host against ~0.02 us here.  Same values, same Montgomery form (R = 2^256).
#pragma once
#include "fp.hpp"

struct H4 { u64 v[4]; };
template <int F> struct HostField {
    static H4 modulus() { H4 p; for (int i = 0; i < 4; ++i) p.v[i] = (u64)FieldParams<F>::P(2 * i) | ((u64)FieldParams<F>::P(2 * i + 1) << 32); return p; }
    static H4 one() { H4 p; for (int i = 0; i < 4; ++i) p.v[i] = (u64)FieldParams<F>::ONE(2 * i) | ((u64)FieldParams<F>::ONE(2 * i + 1) << 32); return p; }
    static u64 inv64() {                 // -p^-1 mod 2^64 by Newton iteration from the 32-bit constant
        const u64 p0 = modulus().v[0];
        u64 x = (u64)(0u - FieldParams<F>::INV32);      // p^-1 mod 2^32
        x *= 2 - p0 * x;                                // mod 2^64
        x *= 2 - p0 * x;
        return 0 - x;
    }
};
static inline bool h4_is_zero(const H4& a) { return (a.v[0] | a.v[1] | a.v[2] | a.v[3]) == 0; }
static inline bool h4_geq(const H4& a, const H4& b) {
    for (int i = 3; i >= 0; --i) { if (a.v[i] != b.v[i]) return a.v[i] > b.v[i]; }
    return true;
}
static inline u64 h4_add_raw(const H4& a, const H4& b, H4& o) {
    unsigned __int128 c = 0;
    for (int i = 0; i < 4; ++i) { c += (unsigned __int128)a.v[i] + b.v[i]; o.v[i] = (u64)c; c >>= 64; }
    return (u64)c;
}
static inline u64 h4_sub_raw(const H4& a, const H4& b, H4& o) {
    u64 br = 0;
    for (int i = 0; i < 4; ++i) { const unsigned __int128 d = (unsigned __int128)a.v[i] - b.v[i] - br; o.v[i] = (u64)d; br = (u64)(d >> 64) & 1; }
    return br;
}
template <int F> static inline H4 h4_add(const H4& a, const H4& b) {
    const H4 p = HostField<F>::modulus();
    H4 s, t;
    const u64 c = h4_add_raw(a, b, s);
    if (c || h4_geq(s, p)) { h4_sub_raw(s, p, t); return t; }
    return s;
}
template <int F> static inline H4 h4_sub(const H4& a, const H4& b) {
    H4 d, t;
    if (h4_sub_raw(a, b, d)) { h4_add_raw(d, HostField<F>::modulus(), t); return t; }
    return d;
}
template <int F> static inline H4 h4_neg(const H4& a) { H4 z = {{0, 0, 0, 0}}; return h4_is_zero(a) ? a : h4_sub<F>(z, a); }
template <int F> static inline H4 h4_dbl(const H4& a) { return h4_add<F>(a, a); }
// CIOS Montgomery product, canonical output
template <int F> static inline H4 h4_mul(const H4& a, const H4& b) {
    static const H4 p = HostField<F>::modulus();
    static const u64 inv = HostField<F>::inv64();
    u64 t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; ++i) {
        unsigned __int128 c = 0;
        for (int j = 0; j < 4; ++j) { c += (unsigned __int128)a.v[j] * b.v[i] + t[j]; t[j] = (u64)c; c >>= 64; }
        c += t[4]; t[4] = (u64)c; t[5] = (u64)(c >> 64);
        const u64 m = t[0] * inv;
        c = (unsigned __int128)m * p.v[0] + t[0]; c >>= 64;
        for (int j = 1; j < 4; ++j) { c += (unsigned __int128)m * p.v[j] + t[j]; t[j - 1] = (u64)c; c >>= 64; }
        c += t[4]; t[3] = (u64)c; t[4] = t[5] + (u64)(c >> 64);
    }
    H4 r = {{t[0], t[1], t[2], t[3]}}, s;
    if (t[4] || h4_geq(r, p)) { h4_sub_raw(r, p, s); return s; }
    return r;
}
