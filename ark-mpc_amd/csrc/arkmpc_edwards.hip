// arkmpc_edwards.hip -- HIP kernels + C ABI for Curve25519 points: the curve group of
// ark_curve25519::EdwardsProjective, which the reference's README example instantiates MpcFabric with (README.md:24).
// Same CurvePoint / PointShare semantics as the BN254 side (online-phase/src/algebra/curve/{curve,share}.rs); only the
// group law differs.
//
// Points cross the ABI as ark-ec twisted-Edwards `Projective{x, y, t, z}` = extended coordinates over Fq = 2^255 - 19 in
// Montgomery form (16 x u64), identity (0, 1, 0, 1); PointShare = 32 x u64.  Results are compared with the oracle on
// affine coordinates.  Curve: -x^2 + y^2 = 1 + d x^2 y^2 (a = -1).  add-2008-hwcd-3 is complete on this curve (a = -1 is
// a square, d is not), so addition needs no exceptional cases at all; doubling is dbl-2008-hwcd.
// Scalars are Curve25519 Fr elements: the context's field must be ARKMPC_CURVE25519_FR.
#include "arkmpc_internal.hpp"
#include "fp_asm.hpp"
#include <cstdlib>
#include <cstring>

#define TPB_ED 128
constexpr int EQ = F_CURVE25519_FQ;
// coordinate multiplications use the hand-scheduled block (tools/gen_asm_kernels.py), as the BN254 point kernels do
#define EQ_MUL(a, b) fe_mul_fast<F_CURVE25519_FQ>(a, b)
#define EQ_SQR(a) EQ_MUL(a, a)
constexpr int ER = F_CURVE25519_FR;
#include "ed25519_consts.inc"

struct Ed {
    Fe x, y, t, z;
};
__device__ __forceinline__ Fe ed_const(const u32 (&c)[8]) {
    Fe r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.v[i] = c[i];
    return r;
}
__device__ __forceinline__ Ed ed_load(const u64* p) {
    Ed r;
    r.x = fe_load(p); r.y = fe_load(p + 4); r.t = fe_load(p + 8); r.z = fe_load(p + 12);
    return r;
}
__device__ __forceinline__ void ed_store(u64* p, const Ed& a) {
    fe_store(p, a.x); fe_store(p + 4, a.y); fe_store(p + 8, a.t); fe_store(p + 12, a.z);
}
__device__ __forceinline__ Ed ed_identity() {
    Ed r;
    r.x = fe_zero<EQ>(); r.y = fe_one<EQ>(); r.t = fe_zero<EQ>(); r.z = fe_one<EQ>();
    return r;
}
__device__ __forceinline__ Ed ed_generator() {
    Ed r;
    r.x = ed_const(ED_GX_MONT); r.y = ed_const(ED_GY_MONT); r.t = ed_const(ED_GT_MONT); r.z = fe_one<EQ>();
    return r;
}
__device__ __forceinline__ Ed ed_select(bool c, const Ed& a, const Ed& b) {
    Ed r;
    r.x = fe_select(c, a.x, b.x); r.y = fe_select(c, a.y, b.y); r.t = fe_select(c, a.t, b.t); r.z = fe_select(c, a.z, b.z);
    return r;
}
__device__ __forceinline__ Ed ed_neg(const Ed& a) {
    Ed r = a;
    r.x = fe_neg<EQ>(a.x); r.t = fe_neg<EQ>(a.t);
    return r;
}
// add-2008-hwcd-3 (a = -1), strongly unified and complete on this curve: 8M + 1 multiplication by 2d
__device__ __noinline__ Ed ed_add(Ed p, Ed q) {
    Fe A = EQ_MUL(fe_sub<EQ>(p.y, p.x), fe_sub<EQ>(q.y, q.x));
    Fe B = EQ_MUL(fe_add<EQ>(p.y, p.x), fe_add<EQ>(q.y, q.x));
    Fe C = EQ_MUL(EQ_MUL(p.t, ed_const(ED_D2_MONT)), q.t);
    Fe D = fe_dbl<EQ>(EQ_MUL(p.z, q.z));
    Fe E = fe_sub<EQ>(B, A), F = fe_sub<EQ>(D, C), G = fe_add<EQ>(D, C), H = fe_add<EQ>(B, A);
    Ed r;
    r.x = EQ_MUL(E, F); r.y = EQ_MUL(G, H); r.t = EQ_MUL(E, H); r.z = EQ_MUL(F, G);
    return r;
}
// dbl-2008-hwcd (a = -1): 4M + 4S
__device__ __noinline__ Ed ed_double(Ed p) {
    Fe A = EQ_SQR(p.x), B = EQ_SQR(p.y), C = fe_dbl<EQ>(EQ_SQR(p.z));
    Fe D = fe_neg<EQ>(A);
    Fe E = fe_sub<EQ>(fe_sub<EQ>(EQ_SQR(fe_add<EQ>(p.x, p.y)), A), B);
    Fe G = fe_add<EQ>(D, B), F = fe_sub<EQ>(G, C), H = fe_sub<EQ>(D, B);
    Ed r;
    r.x = EQ_MUL(E, F); r.y = EQ_MUL(G, H); r.t = EQ_MUL(E, H); r.z = EQ_MUL(F, G);
    return r;
}
// [s]P with 4-bit fixed windows, table k*P (k = 1..15) in the HBM workspace (entry-major), wave-uniform control flow
__device__ __forceinline__ Ed ed_scalar_mul_w4(const Ed& p, const Fe& s_mont, u64* tab, size_t tid, size_t nthreads) {
    const Fe s = fe_to_canonical<ER>(s_mont);
    ed_store(tab + ((size_t)0 * nthreads + tid) * 16, p);
    Ed prev = p;
    for (int k = 2; k <= 15; ++k) {
        Ed t = (k & 1) ? ed_add(prev, p) : ed_double(ed_load(tab + ((size_t)(k / 2 - 1) * nthreads + tid) * 16));
        ed_store(tab + ((size_t)(k - 1) * nthreads + tid) * 16, t);
        prev = t;
    }
    Ed acc = ed_identity();
    for (int limb = 7; limb >= 0; --limb) {
        const u32 w = s.v[limb];
        for (int nib = 7; nib >= 0; --nib) {
            acc = ed_double(ed_double(ed_double(ed_double(acc))));
            const u32 d = (w >> (4 * nib)) & 15u;
            if (__any(d != 0)) {
                Ed q = ed_load(tab + ((size_t)(d ? d - 1 : 0) * nthreads + tid) * 16);
                Ed sum = ed_add(acc, q);
                acc = ed_select(d != 0, sum, acc);
            }
        }
    }
    return acc;
}
// uniform-scalar multiplication without a table (MAC key times a public point)
__device__ __forceinline__ Ed ed_scalar_mul_plain(const Ed& p, const Fe& s_mont) {
    const Fe s = fe_to_canonical<ER>(s_mont);
    Ed acc = ed_identity();
    for (int limb = 7; limb >= 0; --limb) {
        const u32 w = s.v[limb];
        for (int bit = 31; bit >= 0; --bit) {
            acc = ed_double(acc);
            const bool take = (w >> bit) & 1u;
            if (__any(take)) acc = ed_select(take, ed_add(acc, p), acc);
        }
    }
    return acc;
}

// ---------------------------------------------------------------------------------------------
// Variable-base scalar-mul, hand-scheduled form (round 2): prep -> window loop (ed_asm_kernels.inc, tools/gen_ed_asm.py) -> finish.
//   prep    signed 5-bit digits of the canonical scalar (51 windows: |d| <= 16, sign bit) and the table of cached points
//           (Y+X, Y-X, 2dT, 2Z) of 0*P .. 16*P (entry 0 = the identity, so a zero digit needs no special case)
//   loop    one asm stream: accumulator = identity; per window 5 doublings (dbl-2008-hwcd, T only in the last) and one
//           addition of +-T[|d|] (add-2008-hwcd-3: complete on this curve -- no exceptional lanes, no flags, no fallback)
//   finish  values below 2^255 -> canonical, stored as ark-ec's (x, y, t, z)
// ---------------------------------------------------------------------------------------------
#include "ed_asm_kernels.inc"
#include "ed29_asm_kernels.inc"     // the same three streams on nine 29-bit limbs (tools/gen_ed29_asm.py): the default; ARKMPC_ED_LIMBS=32 selects the ones above
static bool ed_limbs29() {
    static const bool on = !(getenv("ARKMPC_ED_LIMBS") && !strcmp(getenv("ARKMPC_ED_LIMBS"), "32"));
    return on;
}
#define TPB_EDLOOP 256
#define ED_ASM_WS_BYTES (ED_ASM_TABLE * 128 + ED_ASM_WINDOWS * 4 + 128)
struct EdAsmWs { u64* tab; u64* res; u32* dig; };
static inline EdAsmWs ed_asm_carve(char* base, size_t n) {
    EdAsmWs w;
    w.tab = (u64*)base; base += n * ED_ASM_TABLE * 128;
    w.res = (u64*)base; base += n * 128;
    w.dig = (u32*)base;
    return w;
}
__device__ __forceinline__ void ed_store_cached(u64* p, const Ed& a) {
    fe_store(p, fe_add<EQ>(a.y, a.x));
    fe_store(p + 4, fe_sub<EQ>(a.y, a.x));
    fe_store(p + 8, EQ_MUL(a.t, ed_const(ED_D2_MONT)));
    fe_store(p + 12, fe_dbl<EQ>(a.z));
}
// signed 5-bit digit records of the canonical scalar, MSB window first (what both prep forms share)
__device__ __forceinline__ void ed_smul_digits(u32 i, u32 n, const u64* scalars, u32 s_stride, u32 s_div, u32* dig) {
    const Fe s = fe_to_canonical<ER>(fe_load(scalars + (size_t)s_stride * (i / s_div)));
    u32 carry = 0;
    for (int j = 0; j < ED_ASM_WINDOWS; ++j) {                 // LSB first; step index = windows - 1 - j (the loop runs MSB first)
        const int bit = 5 * j, limb = bit >> 5, sh = bit & 31;
        u32 lo = 0, hi = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) { lo = (limb == k) ? s.v[k] : lo; hi = (limb + 1 == k) ? s.v[k] : hi; }
        u32 v = lo >> sh;
        if (sh > 27) v |= hi << (32 - sh);
        u32 d = (v & 31u) + carry, neg = 0;
        if (d > 16u) { d = 32u - d; neg = 1; carry = 1; } else carry = 0;
        dig[(size_t)(ED_ASM_WINDOWS - 1 - j) * n + i] = d | ((d ? neg : 0u) << 5);
    }
}
__global__ void __launch_bounds__(256) k_ed_smul_digits(u32 n, const u64* scalars, u32 s_stride, u32 s_div, u32* dig) {
    const u32 i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    ed_smul_digits(i, n, scalars, s_stride, s_div, dig);
}
// the table in the loop's own arithmetic (ed_asm_kernels.inc, ed_smul_table_asm): the arkworks limbs are read as plain field elements --
// the same projective point, see tools/gen_ed_asm.py
__global__ void __launch_bounds__(TPB_EDLOOP) k_ed_smul_table(u32 n, const u64* points, u32 p_stride, u32 p_div, u64* tab) {
    const u32 i = blockIdx.x * TPB_EDLOOP + threadIdx.x;
    if (i >= n) return;
    ed_smul_table_asm(i, p_stride * 8u * (i / p_div), n, points, tab);
}
__global__ void __launch_bounds__(TPB_EDLOOP) k_ed_smul_table29(u32 n, const u64* points, u32 p_stride, u32 p_div, u64* tab) {
    const u32 i = blockIdx.x * TPB_EDLOOP + threadIdx.x;
    if (i >= n) return;
    ed_smul_table29_asm(i, p_stride * 8u * (i / p_div), n, points, tab);
}
__global__ void __launch_bounds__(TPB_EDLOOP) k_ed_smul_loop29(u32 n, u32 tdiv, const u64* tab, const u32* dig, u64* res) {
    const u32 i = blockIdx.x * TPB_EDLOOP + threadIdx.x;
    if (i >= n) return;
    ed_smul_loop29_asm(i, n, i / tdiv, n / tdiv, tab, dig, res);
}
// compiled form of both (generator base, ARKMPC_ED_ASM_PREP=0)
__global__ void __launch_bounds__(TPB_ED) k_ed_smul_prep(u32 n, const u64* points, u32 p_stride, u32 p_div, const u64* scalars, u32 s_stride, u32 s_div,
                                                          EdAsmWs ws) {
    const u32 i = blockIdx.x * TPB_ED + threadIdx.x;
    if (i >= n) return;
    const Ed p = points ? ed_load(points + (size_t)p_stride * (i / p_div)) : ed_generator();
    ed_smul_digits(i, n, scalars, s_stride, s_div, ws.dig);
    ed_store_cached(ws.tab + ((size_t)0 * n + i) * 16, ed_identity());
    Ed acc = p;
    ed_store_cached(ws.tab + ((size_t)1 * n + i) * 16, acc);
    for (int k = 2; k <= 16; ++k) {                            // the unified addition covers k = 2 (P + P) as well
        acc = ed_add(acc, p);
        ed_store_cached(ws.tab + ((size_t)k * n + i) * 16, acc);
    }
}
// tdiv lanes share one table column (ScalarShare x point: the share lane and the MAC lane multiply the same point)
__global__ void __launch_bounds__(TPB_EDLOOP) k_ed_smul_loop(u32 n, u32 tdiv, const u64* tab, const u32* dig, u64* res) {
    const u32 i = blockIdx.x * TPB_EDLOOP + threadIdx.x;
    if (i >= n) return;
    ed_smul_loop_asm(i, n, i / tdiv, n / tdiv, tab, dig, res);
}
__global__ void __launch_bounds__(TPB_EDLOOP) k_ed_smul_finish(u32 n, const u64* res, u64* out) {
    const u32 i = blockIdx.x * TPB_EDLOOP + threadIdx.x;
    if (i >= n) return;
#pragma unroll
    for (int c = 0; c < 4; ++c) fe_store(out + 16 * (size_t)i + 4 * c, fe_reduce_once_loop<EQ>(fe_load(res + 16 * (size_t)i + 4 * c)));
}

template <bool NEGB>
__global__ void __launch_bounds__(TPB_ED) k_ed_add(size_t n, const u64* a, const u64* b, u64* out) {
    size_t i = (size_t)blockIdx.x * TPB_ED + threadIdx.x;
    if (i >= n) return;
    Ed q = ed_load(b + 16 * i);
    if (NEGB) q = ed_neg(q);
    ed_store(out + 16 * i, ed_add(ed_load(a + 16 * i), q));
}
__global__ void __launch_bounds__(TPB_ED) k_ed_neg(size_t n, const u64* a, u64* out) {
    size_t i = (size_t)blockIdx.x * TPB_ED + threadIdx.x;
    if (i >= n) return;
    ed_store(out + 16 * i, ed_neg(ed_load(a + 16 * i)));
}
__global__ void __launch_bounds__(TPB_ED) k_ed_scalar_mul(size_t n, const u64* points, u32 p_stride, u32 p_div, const u64* scalars, u32 s_stride,
                                                           u32 s_div, u64* out, u64* table_ws) {
    size_t i = (size_t)blockIdx.x * TPB_ED + threadIdx.x;
    if (i >= n) return;
    Ed p = points ? ed_load(points + (size_t)p_stride * (i / p_div)) : ed_generator();
    Fe s = fe_load(scalars + (size_t)s_stride * (i / s_div));
    ed_store(out + 16 * i, ed_scalar_mul_w4(p, s, table_ws, i, n));
}
// PointShare::add_public / sub_public (curve/share.rs:57-65): sub_public = add_public(-rhs)
template <bool NEG>
__global__ void __launch_bounds__(TPB_ED) k_edshare_add_public(size_t n, int party, Fe key, const u64* shares, const u64* pub, u64* out) {
    size_t i = (size_t)blockIdx.x * TPB_ED + threadIdx.x;
    if (i >= n) return;
    Ed rhs = ed_load(pub + 16 * i), sh = ed_load(shares + 32 * i), mac = ed_load(shares + 32 * i + 16);
    if (NEG) rhs = ed_neg(rhs);
    if (party == 0) sh = ed_add(sh, rhs);
    mac = ed_add(mac, ed_scalar_mul_plain(rhs, key));
    ed_store(out + 32 * i, sh);
    ed_store(out + 32 * i + 16, mac);
}
// the `.share()` projection of AuthenticatedPointResult::open_batch (authenticated_curve.rs:74-89)
__global__ void __launch_bounds__(TPB_ED) k_edshare_extract(size_t n, const u64* shares, u64* out) {
    size_t i = (size_t)blockIdx.x * TPB_ED + threadIdx.x;
    if (i >= n) return;
    ed_store(out + 16 * i, ed_load(shares + 32 * i));
}
// value * mac_key - share.mac()  (authenticated_curve.rs:215-220)
__global__ void __launch_bounds__(TPB_ED) k_ed_mac_check(size_t n, Fe key, const u64* opened, const u64* shares, u64* out) {
    size_t i = (size_t)blockIdx.x * TPB_ED + threadIdx.x;
    if (i >= n) return;
    Ed v = ed_load(opened + 16 * i), mac = ed_load(shares + 32 * i + 16);
    ed_store(out + 16 * i, ed_add(ed_scalar_mul_plain(v, key), ed_neg(mac)));
}
// my + peer == identity (authenticated_curve.rs:127-131), per element.  Identity in extended coordinates: x = 0 and y = z
// ((0, -1), the point of order 2, has y = -z).
__global__ void __launch_bounds__(TPB_ED) k_ed_mac_verify(size_t n, const u64* mine, const u64* peer, unsigned char* ok) {
    size_t i = (size_t)blockIdx.x * TPB_ED + threadIdx.x;
    if (i >= n) return;
    Ed s = ed_add(ed_load(mine + 16 * i), ed_load(peer + 16 * i));
    ok[i] = (fe_is_zero(s.x) && fe_eq(s.y, s.z)) ? 1 : 0;
}
// Point sums (PointShare::sum, curve/share.rs:85-92; the reduction gate of AuthenticatedPointResult::msm,
// authenticated_curve.rs:796-805): strided per-thread fold, then one LDS tree; blockIdx.y selects the share / MAC column.
#define ED_SUM_TPB 256
__global__ void __launch_bounds__(ED_SUM_TPB) k_ed_partial_sum(size_t n, const u64* pts, u32 stride, u32 lane_off, u64* partial, u32 nthreads) {
    const u32 lane = blockIdx.y, t = blockIdx.x * ED_SUM_TPB + threadIdx.x;
    if (t >= nthreads) return;
    Ed acc = ed_identity();
    for (size_t i = t; i < n; i += nthreads) acc = ed_add(acc, ed_load(pts + (size_t)stride * i + (size_t)lane_off * lane));
    ed_store(partial + ((size_t)lane * nthreads + t) * 16, acc);
}
__global__ void __launch_bounds__(ED_SUM_TPB) k_ed_final_sum(const u64* partial, u32 nthreads, u64* out) {
    __shared__ u64 sm[ED_SUM_TPB * 16];
    const u32 lane = blockIdx.y, t = threadIdx.x;
    Ed acc = ed_identity();
    for (u32 i = t; i < nthreads; i += ED_SUM_TPB) acc = ed_add(acc, ed_load(partial + ((size_t)lane * nthreads + i) * 16));
    ed_store(sm + 16 * t, acc);
    __syncthreads();
    for (u32 s = ED_SUM_TPB / 2; s > 0; s >>= 1) {
        if (t < s) ed_store(sm + 16 * t, ed_add(ed_load(sm + 16 * t), ed_load(sm + 16 * (t + s))));
        __syncthreads();
    }
    if (t == 0) ed_store(out + 16 * lane, ed_load(sm));
}
// z^-1 of a batch of points with Montgomery's trick (z is never 0 on this curve): K points per thread, strided by the thread
// count, one Fermat exponentiation per thread; running products in `pre`
__global__ void __launch_bounds__(TPB_ED) k_ed_zinv(size_t n, u32 K, size_t nthreads, const u64* pts, u64* pre, u64* zinv) {
    const size_t t = (size_t)blockIdx.x * TPB_ED + threadIdx.x;
    if (t >= nthreads) return;
    Fe run = fe_one<EQ>();
    for (u32 j = 0; j < K; ++j) {
        const size_t i = (size_t)j * nthreads + t;
        if (i >= n) break;
        run = EQ_MUL(run, fe_load(pts + 16 * i + 12));
        fe_store(pre + 4 * i, run);
    }
    Fe inv = fe_inv_fermat<EQ>(run);
    for (int j = (int)K - 1; j >= 0; --j) {
        const size_t i = (size_t)j * nthreads + t;
        if (i >= n) continue;
        const Fe before = (j == 0) ? fe_one<EQ>() : fe_load(pre + 4 * ((size_t)(j - 1) * nthreads + t));
        fe_store(zinv + 4 * i, EQ_MUL(inv, before));
        inv = EQ_MUL(inv, fe_load(pts + 16 * i + 12));
    }
}
__global__ void __launch_bounds__(TPB_ED) k_ed_to_affine(size_t n, const u64* pts, const u64* zinv, u64* out_xy) {
    size_t i = (size_t)blockIdx.x * TPB_ED + threadIdx.x;
    if (i >= n) return;
    Ed p = ed_load(pts + 16 * i);
    Fe zi = fe_load(zinv + 4 * i);
    fe_store(out_xy + 8 * i, EQ_MUL(p.x, zi));
    fe_store(out_xy + 8 * i + 4, EQ_MUL(p.y, zi));
}
// ark-serialize compressed twisted-Edwards encoding (CurvePoint::to_bytes, curve.rs:103-108): y little-endian, bit 7 of
// the last byte set iff x > -x (as integers)
__global__ void __launch_bounds__(TPB_ED) k_ed_to_bytes(size_t n, const u64* pts, const u64* zinv, unsigned char* out) {
    size_t i = (size_t)blockIdx.x * TPB_ED + threadIdx.x;
    if (i >= n) return;
    Ed p = ed_load(pts + 16 * i);
    Fe zi = fe_load(zinv + 4 * i);
    Fe x = EQ_MUL(p.x, zi), y = EQ_MUL(p.y, zi);
    Fe xc = fe_to_canonical<EQ>(x), nxc = fe_to_canonical<EQ>(fe_neg<EQ>(x)), yc = fe_to_canonical<EQ>(y);
    u32 br = 0, bo;
#pragma unroll
    for (int k = 0; k < 8; ++k) { (void)__builtin_subc(nxc.v[k], xc.v[k], br, &bo); br = bo; }   // borrow <=> x > -x
    u32 top = yc.v[7];
    if (br) top |= 0x80000000u;
    uint4* q = reinterpret_cast<uint4*>(out + 32 * i);
    q[0] = make_uint4(yc.v[0], yc.v[1], yc.v[2], yc.v[3]);
    q[1] = make_uint4(yc.v[4], yc.v[5], yc.v[6], top);
}

// K9 on this curve: out_i = from_be_bytes_mod_order(SHA3-256(to_bytes(P_i) || to_bytes_be(blinder_i))) with the twisted-Edwards
// compressed encoding (authenticated_curve.rs:227 -> commitment.rs:58-89).  One Keccak-f[1600] per element, all on the GPU.
#include "keccak_device.inc"
__global__ void __launch_bounds__(TPB_ED) k_ed_commit_points(size_t n, const u64* pts, const u64* zinv, const u64* blinders, u64* out) {
    size_t i = (size_t)blockIdx.x * TPB_ED + threadIdx.x;
    if (i >= n) return;
    Ed p = ed_load(pts + 16 * i);
    Fe zi = fe_load(zinv + 4 * i);
    Fe x = EQ_MUL(p.x, zi), y = EQ_MUL(p.y, zi);
    Fe xc = fe_to_canonical<EQ>(x), nxc = fe_to_canonical<EQ>(fe_neg<EQ>(x)), yc = fe_to_canonical<EQ>(y);
    u32 br = 0, bo;
#pragma unroll
    for (int k = 0; k < 8; ++k) { (void)__builtin_subc(nxc.v[k], xc.v[k], br, &bo); br = bo; }   // borrow <=> x > -x
    if (br) yc.v[7] |= 0x80000000u;
    Fe bc = fe_to_canonical<ER>(fe_load(blinders + 4 * i));
    u64 a[25];
#pragma unroll
    for (int k = 0; k < 25; ++k) a[k] = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        a[k] = (u64)yc.v[2 * k] | ((u64)yc.v[2 * k + 1] << 32);
        a[4 + k] = __builtin_bswap64((u64)bc.v[2 * (3 - k)] | ((u64)bc.v[2 * (3 - k) + 1] << 32));
    }
    a[8] ^= 0x06ULL;                    // SHA3 domain separation + first pad bit at byte 64
    a[16] ^= 0x8000000000000000ULL;     // last pad bit at byte 135 (rate = 136)
    keccak_f1600_dev(a);
    Fe v;
#pragma unroll
    for (int k = 0; k < 4; ++k) { u64 w = __builtin_bswap64(a[3 - k]); v.v[2 * k] = (u32)w; v.v[2 * k + 1] = (u32)(w >> 32); }
    fe_store(out + 4 * i, fe_from_canonical<ER>(fe_reduce_once_loop<ER>(v)));
}

// CurvePoint::from_bytes on this curve (curve.rs:110-114 -> ark-ec twisted-Edwards deserialize_compressed with validation),
// the inverse of k_ed_to_bytes: y = low 255 bits (< q), bit 255 selects the larger root x of x^2 = (y^2 - 1) / (d y^2 + 1)
// (q = 5 mod 8: w^((q+3)/8), times sqrt(-1) when that squares to -w), and the point must be in the prime-order subgroup
// ([l]P = O).  ok[i] = 0 and the identity for anything else.
template <int NB> __device__ __forceinline__ Fe eq_pow(const Fe& base, const u32 (&e)[NB]) {
    Fe acc = fe_one<EQ>();
    for (int limb = NB - 1; limb >= 0; --limb) {
        u32 w = 0;
#pragma unroll
        for (int k = 0; k < NB; ++k) w = (limb == k) ? e[k] : w;
        for (int bit = 31; bit >= 0; --bit) {
            acc = EQ_SQR(acc);
            if ((w >> bit) & 1u) acc = EQ_MUL(acc, base);
        }
    }
    return acc;
}
// CHECK: the prime-order subgroup test inside this kernel (253 compiled doublings per point); the hand-scheduled path runs the kernel
// with CHECK = false and verifies [l - 1] P == -P through the scalar-mul pipeline instead (k_ed_subgroup_verify)
template <bool CHECK>
__global__ void __launch_bounds__(TPB_ED) k_ed_from_bytes(size_t n, const unsigned char* in, u64* out, unsigned char* ok) {
    size_t i = (size_t)blockIdx.x * TPB_ED + threadIdx.x;
    if (i >= n) return;
    using PQ = FieldParams<EQ>;
    using PR = FieldParams<ER>;
    const uint4* qv = reinterpret_cast<const uint4*>(in + 32 * i);
    const uint4 a = qv[0], b = qv[1];
    Fe yc;
    yc.v[0] = a.x; yc.v[1] = a.y; yc.v[2] = a.z; yc.v[3] = a.w; yc.v[4] = b.x; yc.v[5] = b.y; yc.v[6] = b.z; yc.v[7] = b.w & 0x7fffffffu;
    const bool flag = (b.w >> 31) & 1u;
    u32 br = 0, bo;
#pragma unroll
    for (int k = 0; k < 8; ++k) { (void)__builtin_subc(yc.v[k], PQ::P(k), br, &bo); br = bo; }
    bool valid = br != 0;                               // y < q
    Ed r = ed_identity();
    if (valid) {
        const Fe y = fe_from_canonical<EQ>(yc);
        const Fe yy = EQ_SQR(y), one = fe_one<EQ>();
        const Fe u = fe_sub<EQ>(yy, one), v = fe_add<EQ>(EQ_MUL(ed_const(ED_D_MONT), yy), one);
        // x = sqrt(u / v) with ONE exponentiation and no inversion (RFC 8032 section 5.1.3): x = u v^3 (u v^7)^((q - 5) / 8), then v x^2 == +-u
        const Fe v2 = EQ_SQR(v), v3 = EQ_MUL(v2, v), v7 = EQ_MUL(EQ_SQR(v3), v);
        u32 e[8], bw = 5;                               // (q - 5) / 8
#pragma unroll
        for (int k = 0; k < 8; ++k) { const u32 pk = PQ::P(k); e[k] = pk - bw; bw = pk < bw ? 1u : 0u; }
#pragma unroll
        for (int k = 0; k < 8; ++k) e[k] = (e[k] >> 3) | (k < 7 ? e[k + 1] << 29 : 0u);
        Fe x = EQ_MUL(EQ_MUL(u, v3), eq_pow<8>(EQ_MUL(u, v7), e));
        const Fe vxx = EQ_MUL(v, EQ_SQR(x));
        if (!fe_eq(vxx, u)) {
            if (fe_eq(vxx, fe_neg<EQ>(u))) x = EQ_MUL(x, ed_const(ED_SQRT_M1_MONT));
            else valid = false;
        }
        const Fe nx = fe_neg<EQ>(x);
        const Fe xc = fe_to_canonical<EQ>(x), nxc = fe_to_canonical<EQ>(nx);
        u32 b2 = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) { (void)__builtin_subc(nxc.v[k], xc.v[k], b2, &bo); b2 = bo; }     // borrows <=> x > -x
        Ed p;
        p.x = ((b2 != 0) == flag) ? x : nx; p.y = y; p.t = EQ_MUL(p.x, y); p.z = one;
        if (CHECK) {                                    // prime-order subgroup: [l]P == identity (x = 0, y = z)
            Ed acc = ed_identity();
            for (int limb = 7; limb >= 0; --limb) {
                const u32 lw = PR::P(limb);
                for (int bit = 31; bit >= 0; --bit) {
                    acc = ed_double(acc);
                    if ((lw >> bit) & 1u) acc = ed_add(acc, p);       // l is a constant: uniform control flow
                }
            }
            valid = valid && fe_is_zero(acc.x) && fe_eq(acc.y, acc.z);
        }
        if (valid) r = p;
    }
    ed_store(out + 16 * i, r);
    ok[i] = valid ? 1 : 0;
}
// prime-order subgroup test on decoded points (z = 1), given r = [l - 1] P from the scalar-mul pipeline: P is in the subgroup iff [l] P is the
// identity iff r == -P, compared projectively (r.x == -x r.z, r.y == y r.z).  A point that fails becomes the identity with ok = 0.
__global__ void __launch_bounds__(TPB_ED) k_ed_subgroup_verify(size_t n, const u64* r_pts, u64* pts, unsigned char* ok) {
    size_t i = (size_t)blockIdx.x * TPB_ED + threadIdx.x;
    if (i >= n) return;
    if (!ok[i]) return;
    const Ed p = ed_load(pts + 16 * i), r = ed_load(r_pts + 16 * i);
    const bool good = fe_eq(r.x, fe_neg<EQ>(EQ_MUL(p.x, r.z))) && fe_eq(r.y, EQ_MUL(p.y, r.z)) && !fe_is_zero(r.z);
    if (!good) { ed_store(pts + 16 * i, ed_identity()); ok[i] = 0; }
}

// ---------------------------------------------------------------------------------------------
// Fixed-base multiplication by the Curve25519 generator (batch_mul_generator on this curve): the multiples
// d * 2^(8w) * G, w = 0..31, d = 1..255, tabulated once per device in "Niels" form (y + x, y - x, 2d*x*y: 96 B each,
// 765 KiB); [s]G is then 32 lookups and at most 32 additions of 7 multiplications each -- no doublings.
// ---------------------------------------------------------------------------------------------
#define EDG_WINDOWS 32
#define EDG_ENTRIES 255
struct EdNiels { Fe ypx, ymx, t2d; };
__device__ __noinline__ Ed ed_madd_niels(Ed p, EdNiels q) {
    Fe A = EQ_MUL(fe_sub<EQ>(p.y, p.x), q.ymx);
    Fe B = EQ_MUL(fe_add<EQ>(p.y, p.x), q.ypx);
    Fe C = EQ_MUL(p.t, q.t2d);
    Fe D = fe_dbl<EQ>(p.z);
    Fe E = fe_sub<EQ>(B, A), F = fe_sub<EQ>(D, C), G = fe_add<EQ>(D, C), H = fe_add<EQ>(B, A);
    Ed r;
    r.x = EQ_MUL(E, F); r.y = EQ_MUL(G, H); r.t = EQ_MUL(E, H); r.z = EQ_MUL(F, G);
    return r;
}
__global__ void __launch_bounds__(64) k_ed_gen_table_bases(u64* bases) {           // one thread: B_w = 2^(8w) G
    if (blockIdx.x | threadIdx.x) return;
    Ed b = ed_generator();
    for (int w = 0; w < EDG_WINDOWS; ++w) {
        ed_store(bases + 16 * w, b);
        for (int k = 0; k < 8; ++k) b = ed_double(b);
    }
}
__global__ void __launch_bounds__(TPB_ED) k_ed_gen_table_fill(const u64* bases, u64* table) {
    const u32 t = blockIdx.x * TPB_ED + threadIdx.x;
    if (t >= EDG_WINDOWS * EDG_ENTRIES) return;
    const u32 w = t / EDG_ENTRIES, d = t % EDG_ENTRIES + 1;
    const Ed b = ed_load(bases + 16 * w);
    Ed acc = ed_identity();
    for (int bit = 7; bit >= 0; --bit) {
        acc = ed_double(acc);
        if ((d >> bit) & 1u) acc = ed_add(acc, b);
    }
    const Fe zi = fe_inv_fermat<EQ>(acc.z);
    const Fe x = EQ_MUL(acc.x, zi), y = EQ_MUL(acc.y, zi);
    fe_store(table + 12 * (size_t)t, fe_add<EQ>(y, x));
    fe_store(table + 12 * (size_t)t + 4, fe_sub<EQ>(y, x));
    fe_store(table + 12 * (size_t)t + 8, EQ_MUL(EQ_MUL(x, y), ed_const(ED_D2_MONT)));
}
__global__ void __launch_bounds__(TPB_ED) k_ed_generator_mul_fixed(size_t n, const u64* scalars, u32 s_stride, u32 s_div, const u64* table, u64* out) {
    size_t i = (size_t)blockIdx.x * TPB_ED + threadIdx.x;
    if (i >= n) return;
    const Fe s = fe_to_canonical<ER>(fe_load(scalars + (size_t)s_stride * (i / s_div)));
    Ed acc = ed_identity();
#pragma unroll 1
    for (int limb = 0; limb < 8; ++limb) {
        u32 wv = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) wv = (limb == k) ? s.v[k] : wv;
#pragma unroll 1
        for (int by = 0; by < 4; ++by) {
            const u32 d = (wv >> (8 * by)) & 255u;
            if (__any(d != 0)) {
                const u64* q = table + 12 * ((size_t)(4 * limb + by) * EDG_ENTRIES + (d ? d - 1 : 0));
                EdNiels e;
                e.ypx = fe_load(q); e.ymx = fe_load(q + 4); e.t2d = fe_load(q + 8);
                acc = ed_select(d != 0, ed_madd_niels(acc, e), acc);
            }
        }
    }
    ed_store(out + 16 * i, acc);
}

// ---- the same multiplication on the plain-arithmetic asm body (ed_asm_kernels.inc, ed_gen_chain_asm): signed 11-bit digits, 23 window
// positions, table T2[w][k] = Niels form (y+x, y-x, 2dxy) of k * 2^(11w) * B as PLAIN canonical field elements (k = 0 is the identity (1, 1, 0),
// so a zero digit needs no predication; k = 1025 exists for the top window's carry): 23 x 1026 x 96 B = 2.2 MiB, L2-resident.
__global__ void __launch_bounds__(64) k_ed_gen2_bases(u64* bases) {           // one thread: B_w = 2^(11w) B
    if (blockIdx.x | threadIdx.x) return;
    Ed b = ed_generator();
    for (int w = 0; w < ED_GEN_ASM_WINDOWS; ++w) {
        ed_store(bases + 16 * w, b);
        for (int k = 0; k < ED_GEN_ASM_C; ++k) b = ed_double(b);
    }
}
__global__ void __launch_bounds__(TPB_ED) k_ed_gen2_fill(const u64* bases, u64* table) {
    const u32 t = blockIdx.x * TPB_ED + threadIdx.x;
    if (t >= ED_GEN_ASM_WINDOWS * ED_GEN_ASM_ENTRIES) return;
    const u32 w = t / ED_GEN_ASM_ENTRIES, d = t % ED_GEN_ASM_ENTRIES;
    const Ed b = ed_load(bases + 16 * w);
    Ed acc = ed_identity();
    for (int bit = ED_GEN_ASM_C; bit >= 0; --bit) {
        acc = ed_double(acc);
        if ((d >> bit) & 1u) acc = ed_add(acc, b);
    }
    const Fe zi = fe_inv_fermat<EQ>(acc.z);
    const Fe x = EQ_MUL(acc.x, zi), y = EQ_MUL(acc.y, zi);
    // Montgomery form -> the canonical integer itself: the asm chain multiplies plain field elements
    fe_store(table + 12 * (size_t)t, fe_to_canonical<EQ>(fe_add<EQ>(y, x)));
    fe_store(table + 12 * (size_t)t + 4, fe_to_canonical<EQ>(fe_sub<EQ>(y, x)));
    fe_store(table + 12 * (size_t)t + 8, fe_to_canonical<EQ>(EQ_MUL(EQ_MUL(x, y), ed_const(ED_D2_MONT))));
}
// records [w][n]: table entry index | sign << 31, signed 11-bit digits LSB window first (the chain has no doublings: any order works)
__global__ void __launch_bounds__(256) k_ed_gen_digits(u32 n, const u64* scalars, u32 s_stride, u32 s_div, u32* dig) {
    const u32 i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const Fe s = fe_to_canonical<ER>(fe_load(scalars + (size_t)s_stride * (i / s_div)));
    u32 carry = 0;
    for (u32 w = 0; w < ED_GEN_ASM_WINDOWS; ++w) {
        const u32 bit = ED_GEN_ASM_C * w, limb = bit >> 5, sh = bit & 31;
        u32 lo = 0, hi = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) { lo = (limb == (u32)k) ? s.v[k] : lo; hi = (limb + 1 == (u32)k) ? s.v[k] : hi; }
        const u64 both = ((u64)hi << 32) | lo;
        u32 d = ((u32)(both >> sh) & ((1u << ED_GEN_ASM_C) - 1u)) + carry, neg = 0;
        // the top window is never recoded: the scalar is below the group order, so its top digit is at most 1024 + carry = 1025
        if (d > (1u << (ED_GEN_ASM_C - 1)) && w + 1 < ED_GEN_ASM_WINDOWS) { d = (1u << ED_GEN_ASM_C) - d; neg = 1; carry = 1; } else carry = 0;
        dig[(size_t)w * n + i] = (w * ED_GEN_ASM_ENTRIES + d) | ((d ? neg : 0u) << 31);
    }
}
__global__ void __launch_bounds__(TPB_EDLOOP) k_ed_gen_chain(u32 n, const u32* dig, const u64* table, u64* res) {
    const u32 i = blockIdx.x * TPB_EDLOOP + threadIdx.x;
    if (i >= n) return;
    ed_gen_chain_asm(i, n, dig, table, res);
}
__global__ void __launch_bounds__(TPB_EDLOOP) k_ed_gen_chain29(u32 n, const u32* dig, const u64* table, u64* res) {
    const u32 i = blockIdx.x * TPB_EDLOOP + threadIdx.x;
    if (i >= n) return;
    ed_gen_chain29_asm(i, n, dig, table, res);
}
static u64* g_ed_gen2_table[16] = {nullptr};
static std::mutex g_ed_gen_mu;
static u64* g_ed_gen_table[16] = {nullptr};
static int ed_gen_table(arkmpc_ctx* ctx, const u64** out) {
    const int dev = ctx->device;
    if (dev < 0 || dev >= 16) return ark_bad(ctx, "device index");
    std::lock_guard<std::mutex> lk(g_ed_gen_mu);
    if (!g_ed_gen_table[dev]) {
        // both tables are built into locals and published together: a failure half way must not leave g_ed_gen_table set with
        // g_ed_gen2_table null (later calls would skip this block and launch the generator chain on a null table)
        u64 *b1 = nullptr, *b2 = nullptr, *t1 = nullptr, *t2 = nullptr;
        hipError_t e = hipMalloc((void**)&b1, EDG_WINDOWS * 128);
        if (e == hipSuccess) e = hipMalloc((void**)&t1, (size_t)EDG_WINDOWS * EDG_ENTRIES * 96);
        if (e == hipSuccess) e = hipMalloc((void**)&b2, ED_GEN_ASM_WINDOWS * 128);
        if (e == hipSuccess) e = hipMalloc((void**)&t2, (size_t)ED_GEN_ASM_WINDOWS * ED_GEN_ASM_ENTRIES * 96);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(k_ed_gen_table_bases, dim3(1), dim3(64), 0, ctx->stream, b1);
            hipLaunchKernelGGL(k_ed_gen_table_fill, dim3(blocks_for(EDG_WINDOWS * EDG_ENTRIES, TPB_ED)), dim3(TPB_ED), 0, ctx->stream, b1, t1);
            hipLaunchKernelGGL(k_ed_gen2_bases, dim3(1), dim3(64), 0, ctx->stream, b2);
            hipLaunchKernelGGL(k_ed_gen2_fill, dim3(blocks_for(ED_GEN_ASM_WINDOWS * ED_GEN_ASM_ENTRIES, TPB_ED)), dim3(TPB_ED), 0, ctx->stream, b2, t2);
            e = hipGetLastError();
            if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        }
        if (b1) (void)hipFree(b1);
        if (b2) (void)hipFree(b2);
        if (e != hipSuccess) {
            if (t1) (void)hipFree(t1);
            if (t2) (void)hipFree(t2);
            ark_set_err(ctx, std::string("generator tables: ") + hipGetErrorString(e));
            return ARKMPC_ERR_HIP;
        }
        g_ed_gen2_table[dev] = t2;
        g_ed_gen_table[dev] = t1;
    }
    *out = g_ed_gen_table[dev];
    return ARKMPC_OK;
}

static void ed_launch_zinv(arkmpc_ctx* ctx, size_t n, const u64* pts, u64* pre, u64* zinv) {
    size_t k = n >> 15;
    const u32 K = (u32)(k < 8 ? 8 : (k > 64 ? 64 : k));
    const size_t threads = (n + K - 1) / K;
    hipLaunchKernelGGL(k_ed_zinv, dim3(blocks_for(threads, TPB_ED)), dim3(TPB_ED), 0, ctx->stream, n, K, threads, pts, pre, zinv);
}

// the hand-scheduled pipeline over m outputs (out_j = points[j / p_div] * scalars[j / s_div]; s_stride = 0 broadcasts one scalar)
static const size_t ED_ASM_CHUNK = (size_t)1 << 19;
static bool ed_asm_enabled() {
    static const bool on = !(getenv("ARKMPC_ED_ASM") && getenv("ARKMPC_ED_ASM")[0] == '0');
    return on;
}
static inline size_t ed_smul_ws_bytes(size_t m) { return (m < ED_ASM_CHUNK ? m : ED_ASM_CHUNK) * ED_ASM_WS_BYTES + 256; }
static void ed_smul_launch(arkmpc_ctx* ctx, size_t m, const u64* points, u32 p_stride, u32 p_div, const u64* scalars, u32 s_stride, u32 s_div, u64* out,
                           char* wsbase) {
    const size_t achunk = m < ED_ASM_CHUNK ? m : ED_ASM_CHUNK;
    for (size_t lo = 0; lo < m; lo += achunk) {
        const size_t cnt = (m - lo < achunk) ? (m - lo) : achunk;
        const u64* pp = points ? points + (size_t)p_stride * (lo / p_div) : (const u64*)nullptr;
        const u64* sp = scalars + (size_t)s_stride * (lo / s_div);
        const EdAsmWs ws = ed_asm_carve(wsbase, cnt);
        static const bool asm_prep = !(getenv("ARKMPC_ED_ASM_PREP") && getenv("ARKMPC_ED_ASM_PREP")[0] == '0');
        u32 tdiv = 1;
        if (asm_prep && pp && (size_t)p_stride * 8 * (cnt / p_div + 1) < ((size_t)1 << 32)) {
            hipLaunchKernelGGL(k_ed_smul_digits, dim3(blocks_for(cnt, 256)), dim3(256), 0, ctx->stream, (u32)cnt, sp, s_stride, s_div, ws.dig);
            if (p_div > 1 && cnt % p_div == 0) tdiv = p_div;          // p_div lanes multiply the same point: one table column serves them
            const u32 ncol = (u32)(cnt / tdiv);
            if (ed_limbs29()) hipLaunchKernelGGL(k_ed_smul_table29, dim3(blocks_for(ncol, TPB_EDLOOP)), dim3(TPB_EDLOOP), 0, ctx->stream, ncol, pp, p_stride, p_div / tdiv, ws.tab);
            else hipLaunchKernelGGL(k_ed_smul_table, dim3(blocks_for(ncol, TPB_EDLOOP)), dim3(TPB_EDLOOP), 0, ctx->stream, ncol, pp, p_stride, p_div / tdiv, ws.tab);
        } else {
            hipLaunchKernelGGL(k_ed_smul_prep, dim3(blocks_for(cnt, TPB_ED)), dim3(TPB_ED), 0, ctx->stream, (u32)cnt, pp, p_stride, p_div, sp, s_stride, s_div, ws);
        }
        // both loops read the same table format (8 x 32-bit words per coordinate, any value below 2^256), whichever kernel built it
        if (ed_limbs29()) hipLaunchKernelGGL(k_ed_smul_loop29, dim3(blocks_for(cnt, TPB_EDLOOP)), dim3(TPB_EDLOOP), 0, ctx->stream, (u32)cnt, tdiv, ws.tab, ws.dig, ws.res);
        else hipLaunchKernelGGL(k_ed_smul_loop, dim3(blocks_for(cnt, TPB_EDLOOP)), dim3(TPB_EDLOOP), 0, ctx->stream, (u32)cnt, tdiv, ws.tab, ws.dig, ws.res);
        hipLaunchKernelGGL(k_ed_smul_finish, dim3(blocks_for(cnt, TPB_EDLOOP)), dim3(TPB_EDLOOP), 0, ctx->stream, (u32)cnt, ws.res, out + 16 * lo);
    }
}
__global__ void __launch_bounds__(64) k_ed_store_scalar(Fe v, u64* out) {
    if (blockIdx.x | threadIdx.x) return;
    fe_store(out, v);
}
// add_public / sub_public and the point MAC check with mac_key * point already computed by the pipeline (kp): additions only
template <bool NEG>
__global__ void __launch_bounds__(TPB_ED) k_edshare_add_public_kp(size_t n, int party, const u64* shares, const u64* pub, const u64* kp, u64* out) {
    size_t i = (size_t)blockIdx.x * TPB_ED + threadIdx.x;
    if (i >= n) return;
    Ed rhs = ed_load(pub + 16 * i), krhs = ed_load(kp + 16 * i), sh = ed_load(shares + 32 * i), mac = ed_load(shares + 32 * i + 16);
    if (NEG) { rhs = ed_neg(rhs); krhs = ed_neg(krhs); }
    if (party == 0) sh = ed_add(sh, rhs);
    mac = ed_add(mac, krhs);
    ed_store(out + 32 * i, sh);
    ed_store(out + 32 * i + 16, mac);
}
__global__ void __launch_bounds__(TPB_ED) k_ed_mac_check_kp(size_t n, const u64* kv, const u64* shares, u64* out) {
    size_t i = (size_t)blockIdx.x * TPB_ED + threadIdx.x;
    if (i >= n) return;
    ed_store(out + 16 * i, ed_add(ed_load(kv + 16 * i), ed_neg(ed_load(shares + 32 * i + 16))));
}

#define ENTER_ED(ctx)                                                                           \
    if (!(ctx)) return ARKMPC_ERR_BAD_ARG;                                                      \
    CtxGuard guard__(ctx);                                                                      \
    if (guard__.rc) return guard__.rc;                                                          \
    if ((ctx)->field_id != ARKMPC_CURVE25519_FR) { ark_set_err((ctx), "Curve25519 point ops need a CURVE25519_FR context"); return ARKMPC_ERR_UNSUPPORTED; }

#include "arkmpc_ed_msm.inc"

// test hook (not part of include/arkmpc.h): the unsaturated arithmetic of the MSM kernels on caller-supplied 256-bit integers, so that its limb
// bounds are exercised at the edges (0, 1, q - 1, q, 2^255 - 20, 2^256 - 1 ...) and not only on the coordinates random points happen to have.
// out: 5 results of 8 words per element, canonical residues mod 2^255 - 19: a b | a + b | a - b | (a - b)(a + b) [loose limbs into the product] | a^-1
__global__ void __launch_bounds__(128) k_f9_selftest(size_t n, const u64* a, const u64* b, u64* out) {
    const size_t i = (size_t)blockIdx.x * 128 + threadIdx.x;
    if (i >= n) return;
    const F9 x = f9_from_words(fe_load(a + 4 * i)), y = f9_from_words(fe_load(b + 4 * i));
    fe_store(out + 20 * i, f9_to_words(f9_mul(x, y)));
    fe_store(out + 20 * i + 4, f9_to_words(f9_add(x, y)));
    fe_store(out + 20 * i + 8, f9_to_words(f9_sub(x, f9_norm(y))));
    fe_store(out + 20 * i + 12, f9_to_words(f9_mul(f9_sub(x, f9_norm(y)), f9_add(x, y))));
    fe_store(out + 20 * i + 16, f9_to_words(f9_inv(x)));
}

extern "C" {

int arkmpc_test_f9(arkmpc_ctx* ctx, size_t n, const uint64_t* a, const uint64_t* b, uint64_t* out) {
    ENTER_ED(ctx);
    Stage st(ctx);
    int ia = st.declare_in(a, n * 32), ib = st.declare_in(b, n * 32), io = st.declare_out(out, n * 160);
    if (st.commit()) return st.rc;
    if (n) hipLaunchKernelGGL(k_f9_selftest, dim3(blocks_for(n, 128)), dim3(128), 0, ctx->stream, n, st.in<u64>(ia), st.in<u64>(ib), st.out<u64>(io));
    return st.finish();
}

int arkmpc_ed_msm(arkmpc_ctx* ctx, size_t n, const uint64_t* points, const uint64_t* scalars, uint64_t* out_point) {
    return ed_msm_impl(ctx, n, points, scalars, 4, 1, out_point);
}
int arkmpc_ed_msm_authenticated(arkmpc_ctx* ctx, size_t n, const uint64_t* points, const uint64_t* scalar_shares, uint64_t* out_share) {
    return ed_msm_impl(ctx, n, points, scalar_shares, 8, 2, out_share);
}

static int ed_addsub(arkmpc_ctx* ctx, bool sub, size_t m, const uint64_t* a, const uint64_t* b, uint64_t* out) {
    ENTER_ED(ctx);
    Stage st(ctx);
    int ia = st.declare_in(a, m * 128), ib = st.declare_in(b, m * 128), io = st.declare_out(out, m * 128);
    if (st.commit()) return st.rc;
    if (m) {
        dim3 g(blocks_for(m, TPB_ED)), t(TPB_ED);
        if (sub) hipLaunchKernelGGL((k_ed_add<true>), g, t, 0, ctx->stream, m, st.in<u64>(ia), st.in<u64>(ib), st.out<u64>(io));
        else hipLaunchKernelGGL((k_ed_add<false>), g, t, 0, ctx->stream, m, st.in<u64>(ia), st.in<u64>(ib), st.out<u64>(io));
    }
    return st.finish();
}
int arkmpc_ed_add(arkmpc_ctx* ctx, size_t n, const uint64_t* a, const uint64_t* b, uint64_t* out) { return ed_addsub(ctx, false, n, a, b, out); }
int arkmpc_ed_sub(arkmpc_ctx* ctx, size_t n, const uint64_t* a, const uint64_t* b, uint64_t* out) { return ed_addsub(ctx, true, n, a, b, out); }
int arkmpc_edshare_add(arkmpc_ctx* ctx, size_t n, const uint64_t* a, const uint64_t* b, uint64_t* out) { return ed_addsub(ctx, false, 2 * n, a, b, out); }
int arkmpc_edshare_sub(arkmpc_ctx* ctx, size_t n, const uint64_t* a, const uint64_t* b, uint64_t* out) { return ed_addsub(ctx, true, 2 * n, a, b, out); }

static int ed_neg_impl(arkmpc_ctx* ctx, size_t m, const uint64_t* a, uint64_t* out) {
    ENTER_ED(ctx);
    Stage st(ctx);
    int ia = st.declare_in(a, m * 128), io = st.declare_out(out, m * 128);
    if (st.commit()) return st.rc;
    if (m) hipLaunchKernelGGL(k_ed_neg, dim3(blocks_for(m, TPB_ED)), dim3(TPB_ED), 0, ctx->stream, m, st.in<u64>(ia), st.out<u64>(io));
    return st.finish();
}
int arkmpc_ed_neg(arkmpc_ctx* ctx, size_t n, const uint64_t* a, uint64_t* out) { return ed_neg_impl(ctx, n, a, out); }
int arkmpc_edshare_neg(arkmpc_ctx* ctx, size_t n, const uint64_t* a, uint64_t* out) { return ed_neg_impl(ctx, 2 * n, a, out); }

static int ed_smul_impl(arkmpc_ctx* ctx, size_t m, const uint64_t* points, size_t n_points, u32 p_stride, u32 p_div, const uint64_t* scalars,
                        size_t scalar_bytes, u32 s_stride, u32 s_div, uint64_t* out) {
    ENTER_ED(ctx);
    Stage st(ctx);
    int ip = points ? st.declare_in(points, n_points * 128) : -1;
    int is = st.declare_in(scalars, scalar_bytes), io = st.declare_out(out, m * 128);
    const size_t CH = (size_t)1 << 20;
    const size_t chunk = m < CH ? m : CH;
    static const bool fixed_base = !(getenv("ARKMPC_NO_FIXED_BASE") && getenv("ARKMPC_NO_FIXED_BASE")[0] == '1');
    const bool asm_loop = ed_asm_enabled();
    int iw = -1, igd = -1, igr = -1;
    if (points || !fixed_base) iw = asm_loop ? st.declare_scratch(ed_smul_ws_bytes(m)) : st.declare_scratch(chunk * 15 * 128);
    else if (asm_loop) { igd = st.declare_scratch(chunk * ED_GEN_ASM_WINDOWS * 4 + 64); igr = st.declare_scratch(chunk * 128 + 64); }
    if (st.commit()) return st.rc;
    if (m && !points && fixed_base) {
        const u64* table = nullptr;
        int rc = ed_gen_table(ctx, &table);
        if (rc) return rc;
        if (asm_loop) {                                    // signed 11-bit digits, 23 additions on the plain-arithmetic asm body
            const u64* t2 = g_ed_gen2_table[ctx->device];
            for (size_t lo = 0; lo < m; lo += chunk) {
                const size_t cnt = (m - lo < chunk) ? (m - lo) : chunk;
                const u64* sp = st.in<u64>(is) + (size_t)s_stride * (lo / s_div);
                hipLaunchKernelGGL(k_ed_gen_digits, dim3(blocks_for(cnt, 256)), dim3(256), 0, ctx->stream, (u32)cnt, sp, s_stride, s_div, st.scratch<u32>(igd));
                if (ed_limbs29()) hipLaunchKernelGGL(k_ed_gen_chain29, dim3(blocks_for(cnt, TPB_EDLOOP)), dim3(TPB_EDLOOP), 0, ctx->stream, (u32)cnt, st.scratch<u32>(igd), t2, st.scratch<u64>(igr));
                else hipLaunchKernelGGL(k_ed_gen_chain, dim3(blocks_for(cnt, TPB_EDLOOP)), dim3(TPB_EDLOOP), 0, ctx->stream, (u32)cnt, st.scratch<u32>(igd), t2, st.scratch<u64>(igr));
                hipLaunchKernelGGL(k_ed_smul_finish, dim3(blocks_for(cnt, TPB_EDLOOP)), dim3(TPB_EDLOOP), 0, ctx->stream, (u32)cnt, st.scratch<u64>(igr), st.out<u64>(io) + 16 * lo);
            }
            return st.finish();
        }
        hipLaunchKernelGGL(k_ed_generator_mul_fixed, dim3(blocks_for(m, TPB_ED)), dim3(TPB_ED), 0, ctx->stream, m, st.in<u64>(is), s_stride, s_div, table,
                           st.out<u64>(io));
        return st.finish();
    }
    if (m && asm_loop) {                                   // hand-scheduled window loop (prep / loop / finish kernels)
        ed_smul_launch(ctx, m, points ? st.in<u64>(ip) : (const u64*)nullptr, p_stride, p_div, st.in<u64>(is), s_stride, s_div, st.out<u64>(io), st.scratch<char>(iw));
        return st.finish();
    }
    for (size_t lo = 0; lo < m; lo += chunk) {
        const size_t cnt = (m - lo < chunk) ? (m - lo) : chunk;
        const u64* pp = points ? st.in<u64>(ip) + (size_t)p_stride * (lo / p_div) : (const u64*)nullptr;
        const u64* sp = st.in<u64>(is) + (size_t)s_stride * (lo / s_div);
        hipLaunchKernelGGL(k_ed_scalar_mul, dim3(blocks_for(cnt, TPB_ED)), dim3(TPB_ED), 0, ctx->stream, cnt, pp, p_stride, p_div, sp, s_stride,
                           s_div, st.out<u64>(io) + 16 * lo, st.scratch<u64>(iw));
    }
    return st.finish();
}
int arkmpc_ed_scalar_mul(arkmpc_ctx* ctx, size_t n, const uint64_t* points, const uint64_t* scalars, uint64_t* out) {
    if (n && !points) return ctx ? ark_bad(ctx, "null points") : ARKMPC_ERR_BAD_ARG;
    return ed_smul_impl(ctx, n, points, n, 16, 1, scalars, n * 32, 4, 1, out);
}
int arkmpc_ed_generator_mul(arkmpc_ctx* ctx, size_t n, const uint64_t* scalars, uint64_t* out) {
    return ed_smul_impl(ctx, n, nullptr, 0, 0, 1, scalars, n * 32, 4, 1, out);
}
int arkmpc_edshare_mul_public(arkmpc_ctx* ctx, size_t n, const uint64_t* shares, const uint64_t* scalars, uint64_t* out) {
    if (n && !shares) return ctx ? ark_bad(ctx, "null shares") : ARKMPC_ERR_BAD_ARG;
    return ed_smul_impl(ctx, 2 * n, shares, 2 * n, 16, 1, scalars, n * 32, 4, 2, out);
}
int arkmpc_scalarshare_mul_ed_generator(arkmpc_ctx* ctx, size_t n, const uint64_t* scalar_shares, uint64_t* out) {
    return ed_smul_impl(ctx, 2 * n, nullptr, 0, 0, 1, scalar_shares, n * 64, 4, 1, out);
}
static int edshare_addsub_public(arkmpc_ctx* ctx, bool sub, size_t n, int party_id, const uint64_t mac_key[4], const uint64_t* shares,
                                 const uint64_t* pub_points, uint64_t* out) {
    ENTER_ED(ctx);
    if (party_id != 0 && party_id != 1) return ark_bad(ctx, "party_id must be 0 or 1");
    if (!mac_key) return ark_bad(ctx, "null mac_key");
    Stage st(ctx);
    int is = st.declare_in(shares, n * 256), ip = st.declare_in(pub_points, n * 128), io = st.declare_out(out, n * 256);
    const bool asm_loop = ed_asm_enabled();
    int iw = asm_loop ? st.declare_scratch(ed_smul_ws_bytes(n)) : -1, ik = asm_loop ? st.declare_scratch(n * 128 + 64) : -1;
    if (st.commit()) return st.rc;
    const dim3 g(blocks_for(n, TPB_ED)), t(TPB_ED);
    if (n && asm_loop) {                                   // mac_key * rhs through the scalar-mul pipeline (one broadcast scalar)
        u64* dkey = st.scratch<u64>(ik);
        u64* kp = dkey + 8;
        hipLaunchKernelGGL(k_ed_store_scalar, dim3(1), dim3(64), 0, ctx->stream, fe_from_host(mac_key), dkey);
        ed_smul_launch(ctx, n, st.in<u64>(ip), 16, 1, dkey, 0, 1, kp, st.scratch<char>(iw));
        if (sub) hipLaunchKernelGGL(k_edshare_add_public_kp<true>, g, t, 0, ctx->stream, n, party_id, st.in<u64>(is), st.in<u64>(ip), kp, st.out<u64>(io));
        else hipLaunchKernelGGL(k_edshare_add_public_kp<false>, g, t, 0, ctx->stream, n, party_id, st.in<u64>(is), st.in<u64>(ip), kp, st.out<u64>(io));
        return st.finish();
    }
    if (n && sub) hipLaunchKernelGGL(k_edshare_add_public<true>, g, t, 0, ctx->stream, n, party_id, fe_from_host(mac_key), st.in<u64>(is), st.in<u64>(ip), st.out<u64>(io));
    else if (n) hipLaunchKernelGGL(k_edshare_add_public<false>, g, t, 0, ctx->stream, n, party_id, fe_from_host(mac_key), st.in<u64>(is), st.in<u64>(ip), st.out<u64>(io));
    return st.finish();
}
int arkmpc_edshare_add_public(arkmpc_ctx* ctx, size_t n, int party_id, const uint64_t mac_key[4], const uint64_t* shares,
                              const uint64_t* pub_points, uint64_t* out) {
    return edshare_addsub_public(ctx, false, n, party_id, mac_key, shares, pub_points, out);
}
int arkmpc_edshare_sub_public(arkmpc_ctx* ctx, size_t n, int party_id, const uint64_t mac_key[4], const uint64_t* shares,
                              const uint64_t* pub_points, uint64_t* out) {
    return edshare_addsub_public(ctx, true, n, party_id, mac_key, shares, pub_points, out);
}
// ScalarShare * CurvePoint on this curve (curve.rs:483-517): output point j = points[j / 2] * scalar_shares_as_scalars[j]
int arkmpc_scalarshare_mul_ed_point(arkmpc_ctx* ctx, size_t n, const uint64_t* scalar_shares, const uint64_t* points, uint64_t* out) {
    if (n && !points) return ctx ? ark_bad(ctx, "null points") : ARKMPC_ERR_BAD_ARG;
    return ed_smul_impl(ctx, 2 * n, points, n, 16, 2, scalar_shares, n * 64, 4, 1, out);
}
int arkmpc_edshare_extract(arkmpc_ctx* ctx, size_t n, const uint64_t* shares, uint64_t* out_points) {
    ENTER_ED(ctx);
    Stage st(ctx);
    int is = st.declare_in(shares, n * 256), io = st.declare_out(out_points, n * 128);
    if (st.commit()) return st.rc;
    if (n) hipLaunchKernelGGL(k_edshare_extract, dim3(blocks_for(n, TPB_ED)), dim3(TPB_ED), 0, ctx->stream, n, st.in<u64>(is), st.out<u64>(io));
    return st.finish();
}
int arkmpc_ed_mac_check_shares(arkmpc_ctx* ctx, size_t n, const uint64_t mac_key[4], const uint64_t* opened_points, const uint64_t* shares,
                               uint64_t* out_chk_points) {
    ENTER_ED(ctx);
    if (!mac_key) return ark_bad(ctx, "null mac_key");
    Stage st(ctx);
    int iv = st.declare_in(opened_points, n * 128), is = st.declare_in(shares, n * 256), io = st.declare_out(out_chk_points, n * 128);
    const bool asm_loop = ed_asm_enabled();
    int iw = asm_loop ? st.declare_scratch(ed_smul_ws_bytes(n)) : -1, ik = asm_loop ? st.declare_scratch(n * 128 + 64) : -1;
    if (st.commit()) return st.rc;
    if (n && asm_loop) {
        u64* dkey = st.scratch<u64>(ik);
        u64* kv = dkey + 8;
        hipLaunchKernelGGL(k_ed_store_scalar, dim3(1), dim3(64), 0, ctx->stream, fe_from_host(mac_key), dkey);
        ed_smul_launch(ctx, n, st.in<u64>(iv), 16, 1, dkey, 0, 1, kv, st.scratch<char>(iw));
        hipLaunchKernelGGL(k_ed_mac_check_kp, dim3(blocks_for(n, TPB_ED)), dim3(TPB_ED), 0, ctx->stream, n, kv, st.in<u64>(is), st.out<u64>(io));
        return st.finish();
    }
    if (n) hipLaunchKernelGGL(k_ed_mac_check, dim3(blocks_for(n, TPB_ED)), dim3(TPB_ED), 0, ctx->stream, n, fe_from_host(mac_key), st.in<u64>(iv),
                              st.in<u64>(is), st.out<u64>(io));
    return st.finish();
}
int arkmpc_ed_mac_verify(arkmpc_ctx* ctx, size_t n, const uint64_t* mine, const uint64_t* peer, uint8_t* out_ok) {
    ENTER_ED(ctx);
    Stage st(ctx);
    int im = st.declare_in(mine, n * 128), ip = st.declare_in(peer, n * 128), io = st.declare_out(out_ok, n);
    if (st.commit()) return st.rc;
    if (n) hipLaunchKernelGGL(k_ed_mac_verify, dim3(blocks_for(n, TPB_ED)), dim3(TPB_ED), 0, ctx->stream, n, st.in<u64>(im), st.in<u64>(ip),
                              st.out<unsigned char>(io));
    return st.finish();
}
int arkmpc_commit_ed_points_sha3(arkmpc_ctx* ctx, size_t n, const uint64_t* points, const uint64_t* blinders, uint64_t* out_commitments) {
    ENTER_ED(ctx);
    Stage st(ctx);
    int ip = st.declare_in(points, n * 128), ib = st.declare_in(blinders, n * 32), io = st.declare_out(out_commitments, n * 32);
    int iz = st.declare_scratch(n * 64 + 64);
    if (st.commit()) return st.rc;
    if (n) {
        ed_launch_zinv(ctx, n, st.in<u64>(ip), st.scratch<u64>(iz), st.scratch<u64>(iz) + 4 * n);
        hipLaunchKernelGGL(k_ed_commit_points, dim3(blocks_for(n, TPB_ED)), dim3(TPB_ED), 0, ctx->stream, n, st.in<u64>(ip), st.scratch<u64>(iz) + 4 * n,
                           st.in<u64>(ib), st.out<u64>(io));
    }
    return st.finish();
}
static int ed_sum_impl(arkmpc_ctx* ctx, size_t n, const uint64_t* pts, u32 stride, u32 lanes, uint64_t* out) {
    ENTER_ED(ctx);
    Stage st(ctx);
    int ip = st.declare_in(pts, n * stride * 8), io = st.declare_out(out, (size_t)lanes * 128);
    // three short levels instead of one long one: every thread of a level folds at most ~16 inputs (a fold is a DEPENDENT chain of
    // additions, ~5 us each), so 2^18 points take ~40 dependent additions instead of ~270 (2.98 -> 0.3 ms on BN254)
    size_t t1 = n / 8; t1 = t1 < 256 ? (n ? (n < 256 ? n : 256) : 1) : (t1 > 65536 ? 65536 : t1);
    const u32 n1 = (u32)t1, n2 = n1 > 2048 ? n1 / 16 : 0;
    int iw = st.declare_scratch((size_t)lanes * n1 * 128), iw2 = st.declare_scratch((size_t)lanes * (n2 ? n2 : 1) * 128);
    if (st.commit()) return st.rc;
    hipLaunchKernelGGL(k_ed_partial_sum, dim3(blocks_for(n1, ED_SUM_TPB), lanes), dim3(ED_SUM_TPB), 0, ctx->stream, n, st.in<u64>(ip), stride, 16u,
                       st.scratch<u64>(iw), n1);
    const u64* last = st.scratch<u64>(iw);
    u32 nlast = n1;
    if (n2) {                                               // level 2 reads level 1's per-lane arrays: stride = one point, lane offset = one array
        hipLaunchKernelGGL(k_ed_partial_sum, dim3(blocks_for(n2, ED_SUM_TPB), lanes), dim3(ED_SUM_TPB), 0, ctx->stream, (size_t)n1, last, 16u, n1 * 16u,
                           st.scratch<u64>(iw2), n2);
        last = st.scratch<u64>(iw2); nlast = n2;
    }
    hipLaunchKernelGGL(k_ed_final_sum, dim3(1, lanes), dim3(ED_SUM_TPB), 0, ctx->stream, last, nlast, st.out<u64>(io));
    return st.finish();
}
int arkmpc_ed_sum(arkmpc_ctx* ctx, size_t n, const uint64_t* points, uint64_t* out_point) { return ed_sum_impl(ctx, n, points, 16, 1, out_point); }
int arkmpc_edshare_sum(arkmpc_ctx* ctx, size_t n, const uint64_t* shares, uint64_t* out_share) { return ed_sum_impl(ctx, n, shares, 32, 2, out_share); }
int arkmpc_ed_to_affine(arkmpc_ctx* ctx, size_t n, const uint64_t* points, uint64_t* out_xy) {
    ENTER_ED(ctx);
    Stage st(ctx);
    int ip = st.declare_in(points, n * 128), io = st.declare_out(out_xy, n * 64), iz = st.declare_scratch(n * 64 + 64);
    if (st.commit()) return st.rc;
    if (n) {
        ed_launch_zinv(ctx, n, st.in<u64>(ip), st.scratch<u64>(iz), st.scratch<u64>(iz) + 4 * n);
        hipLaunchKernelGGL(k_ed_to_affine, dim3(blocks_for(n, TPB_ED)), dim3(TPB_ED), 0, ctx->stream, n, st.in<u64>(ip), st.scratch<u64>(iz) + 4 * n, st.out<u64>(io));
    }
    return st.finish();
}
int arkmpc_ed_from_bytes(arkmpc_ctx* ctx, size_t n, const uint8_t* bytes, uint64_t* out_points, uint8_t* out_ok) {
    ENTER_ED(ctx);
    Stage st(ctx);
    int ib = st.declare_in(bytes, n * 32), io = st.declare_out(out_points, n * 128), ik = st.declare_out(out_ok, n);
    const bool asm_loop = ed_asm_enabled();
    int iw = asm_loop ? st.declare_scratch(ed_smul_ws_bytes(n)) : -1, ir = asm_loop ? st.declare_scratch(n * 128 + 128) : -1;
    if (st.commit()) return st.rc;
    const dim3 g(blocks_for(n, TPB_ED)), t(TPB_ED);
    if (n && asm_loop) {
        // decode without the subgroup loop, then [l - 1] P for all points through the hand-scheduled pipeline (one broadcast scalar) and compare with -P
        hipLaunchKernelGGL(k_ed_from_bytes<false>, g, t, 0, ctx->stream, n, st.in<unsigned char>(ib), st.out<u64>(io), st.out<unsigned char>(ik));
        u64* dkey = st.scratch<u64>(ir);
        u64* rp = dkey + 16;
        hipLaunchKernelGGL(k_ed_store_scalar, dim3(1), dim3(64), 0, ctx->stream, fe_neg<ER>(fe_one<ER>()), dkey);       // l - 1 in Montgomery form
        ed_smul_launch(ctx, n, st.out<u64>(io), 16, 1, dkey, 0, 1, rp, st.scratch<char>(iw));
        hipLaunchKernelGGL(k_ed_subgroup_verify, g, t, 0, ctx->stream, n, rp, st.out<u64>(io), st.out<unsigned char>(ik));
    } else if (n) {
        hipLaunchKernelGGL(k_ed_from_bytes<true>, g, t, 0, ctx->stream, n, st.in<unsigned char>(ib), st.out<u64>(io), st.out<unsigned char>(ik));
    }
    return st.finish();
}
int arkmpc_ed_to_bytes(arkmpc_ctx* ctx, size_t n, const uint64_t* points, uint8_t* out_bytes) {
    ENTER_ED(ctx);
    Stage st(ctx);
    int ip = st.declare_in(points, n * 128), io = st.declare_out(out_bytes, n * 32), iz = st.declare_scratch(n * 64 + 64);
    if (st.commit()) return st.rc;
    if (n) {
        ed_launch_zinv(ctx, n, st.in<u64>(ip), st.scratch<u64>(iz), st.scratch<u64>(iz) + 4 * n);
        hipLaunchKernelGGL(k_ed_to_bytes, dim3(blocks_for(n, TPB_ED)), dim3(TPB_ED), 0, ctx->stream, n, st.in<u64>(ip), st.scratch<u64>(iz) + 4 * n,
                           st.out<unsigned char>(io));
    }
    return st.finish();
}

}  // extern "C"
