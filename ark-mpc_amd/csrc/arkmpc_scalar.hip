// arkmpc_scalar.hip -- HIP kernels + C ABI for the scalar side of ark-mpc's authenticated-share
// hot path: Scalar / ScalarShare batch ops, Beaver
// multiplication (K1 mask, K2 combine, K3 finish), batch open + MAC check (K4, K5, K6, H1).
//
// Kernel shape: one thread per gate, 256-thread workgroups (4 wave64), grid = ceil(n/256) -- at
// n = 2^20 that is 4096 workgroups over 256 CUs.  Gates are independent, so there is no LDS use
// and no inter-workgroup traffic; every kernel is a pure stream over HBM.  Elements are read as
// 16-byte vector loads (two per 32-byte field element); measured on MI355X the 64-byte-strided
// AoS pattern reaches 96% of a planar layout's rate (profiles/ubench_r01.log), so the arkworks
// AoS layout is consumed directly, with no transpose pass.
#include "arkmpc_internal.hpp"
#include <cstdlib>

#define TPB 256

// A column of a ScalarShare vector: element i lives at base + i*stride (u64 units).
struct Col {
    const u64* p;
    u32 stride;
};
struct ColOut {
    u64* p;
    u32 stride;
};

#include "fp_asm.hpp"

// ---------------------------------------------------------------------------------------------
// elementwise Scalar kernels
// ---------------------------------------------------------------------------------------------
enum { OP_ADD = 0, OP_SUB = 1, OP_MUL = 2 };

template <int F, int OP>
__global__ void __launch_bounds__(TPB) k_scalar_binop(size_t n, const u64* a, const u64* b, u64* out) {
    size_t i = (size_t)blockIdx.x * TPB + threadIdx.x;
    if (i >= n) return;
    Fe x = fe_load(a + 4 * i), y = fe_load(b + 4 * i), r;
    if (OP == OP_ADD) r = fe_add<F>(x, y);
    if (OP == OP_SUB) r = fe_sub<F>(x, y);
    if (OP == OP_MUL) r = fe_mul<F>(x, y);
    fe_store(out + 4 * i, r);
}
template <int F>
__global__ void __launch_bounds__(TPB) k_scalar_neg(size_t n, const u64* a, u64* out) {
    size_t i = (size_t)blockIdx.x * TPB + threadIdx.x;
    if (i >= n) return;
    fe_store(out + 4 * i, fe_neg<F>(fe_load(a + 4 * i)));
}
template <int F>
__global__ void __launch_bounds__(TPB) k_from_canonical(size_t n, const u64* a, u64* out) {
    size_t i = (size_t)blockIdx.x * TPB + threadIdx.x;
    if (i >= n) return;
    fe_store(out + 4 * i, fe_from_canonical<F>(fe_reduce_once_loop<F>(fe_load(a + 4 * i))));
}
template <int F>
__global__ void __launch_bounds__(TPB) k_to_canonical(size_t n, const u64* a, u64* out) {
    size_t i = (size_t)blockIdx.x * TPB + threadIdx.x;
    if (i >= n) return;
    fe_store(out + 4 * i, fe_to_canonical<F>(fe_load(a + 4 * i)));
}
// K6: Scalar::to_bytes_be (scalar.rs:118-127).  32 big-endian bytes = limbs reversed, bytes swapped.
template <int F>
__global__ void __launch_bounds__(TPB) k_to_bytes_be(size_t n, const u64* a, unsigned char* out) {
    size_t i = (size_t)blockIdx.x * TPB + threadIdx.x;
    if (i >= n) return;
    Fe c = fe_to_canonical<F>(fe_load(a + 4 * i));
    uint4* q = reinterpret_cast<uint4*>(out + 32 * i);
    q[0] = make_uint4(__builtin_bswap32(c.v[7]), __builtin_bswap32(c.v[6]), __builtin_bswap32(c.v[5]), __builtin_bswap32(c.v[4]));
    q[1] = make_uint4(__builtin_bswap32(c.v[3]), __builtin_bswap32(c.v[2]), __builtin_bswap32(c.v[1]), __builtin_bswap32(c.v[0]));
}

// Scalar::batch_inverse (scalar.rs:93-100 -> ark_ff::batch_inversion): every non-zero element is replaced by its
// inverse, zeros stay zero.  Montgomery's trick around ONE Fermat exponentiation (a^(p-2), ~380 multiplications) per thread:
// a thread owns K elements strided by the thread count (coalesced), writes their running products to a scratch column,
// inverts the total, and walks back: 3 multiplications per element + 380 / K.  K grows with the batch (8 .. 64): at 2^22
// elements 9 multiplications per element instead of the 50 of a fixed K = 8 (1.97 ms -> see tools/kernel_suite.py).
template <int F>
__global__ void __launch_bounds__(TPB) k_batch_inverse(size_t n, u32 K, size_t nthreads, const u64* a, u64* pre, u64* out) {
    const size_t t = (size_t)blockIdx.x * TPB + threadIdx.x;
    if (t >= nthreads) return;
    Fe run = fe_one<F>();
    for (u32 j = 0; j < K; ++j) {
        const size_t i = (size_t)j * nthreads + t;
        if (i >= n) break;
        const Fe v = fe_load(a + 4 * i);
        if (!fe_is_zero(v)) run = fe_mul<F>(run, v);
        fe_store(pre + 4 * i, run);                 // product of the non-zero elements of this thread up to and including j
    }
    Fe inv = fe_inv_fermat<F>(run);
    for (int j = (int)K - 1; j >= 0; --j) {
        const size_t i = (size_t)j * nthreads + t;
        if (i >= n) continue;
        const Fe v = fe_load(a + 4 * i);            // read before out[i] is written: in-place calls are fine
        Fe o = fe_zero<F>();
        if (!fe_is_zero(v)) {
            const Fe before = (j == 0) ? fe_one<F>() : fe_load(pre + 4 * ((size_t)(j - 1) * nthreads + t));
            o = fe_mul<F>(inv, before);
            inv = fe_mul<F>(inv, v);
        }
        fe_store(out + 4 * i, o);
    }
}

// Inclusive prefix product out_i = in_0 * ... * in_i (the public scan inside gadgets.rs:131-137 prefix_product, there a
// sequential loop of ScalarResult multiplications).  Field multiplication is associative, so it is a parallel scan:
// every thread folds SCAN_ITEMS consecutive elements, the workgroup scans the 256 thread totals through LDS
// (Hillis-Steele, 8 steps), block totals are scanned recursively and applied.
#define SCAN_ITEMS 8
#define SCAN_BLOCK (TPB * SCAN_ITEMS)
template <int F>
__global__ void __launch_bounds__(TPB) k_scan_block(size_t n, const u64* in, u64* out, u64* block_totals) {
    __shared__ u64 sm[TPB * 4];
    const size_t base = ((size_t)blockIdx.x * TPB + threadIdx.x) * SCAN_ITEMS;
    Fe v[SCAN_ITEMS];
    Fe run = fe_one<F>();
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        v[i] = (base + i < n) ? fe_load(in + 4 * (base + i)) : fe_one<F>();
        run = fe_mul<F>(run, v[i]);
        v[i] = run;                                        // thread-local inclusive prefix
    }
    fe_store(sm + 4 * threadIdx.x, run);
    __syncthreads();
    for (int off = 1; off < TPB; off <<= 1) {              // inclusive scan of the thread totals
        Fe mine = fe_load(sm + 4 * threadIdx.x), left = fe_one<F>();
        const bool has = threadIdx.x >= (unsigned)off;
        if (has) left = fe_load(sm + 4 * (threadIdx.x - off));
        __syncthreads();
        if (has) fe_store(sm + 4 * threadIdx.x, fe_mul<F>(left, mine));
        __syncthreads();
    }
    const Fe excl = threadIdx.x ? fe_load(sm + 4 * (threadIdx.x - 1)) : fe_one<F>();
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i)
        if (base + i < n) fe_store(out + 4 * (base + i), fe_mul<F>(excl, v[i]));
    if (threadIdx.x == TPB - 1 && block_totals) fe_store(block_totals + 4 * blockIdx.x, fe_load(sm + 4 * (TPB - 1)));
}
template <int F>
__global__ void __launch_bounds__(TPB) k_scan_apply(size_t n, u64* out, const u64* scanned_totals) {
    const size_t i = (size_t)blockIdx.x * TPB + threadIdx.x;
    const size_t blk = i / SCAN_BLOCK;
    if (i >= n || blk == 0) return;
    fe_store(out + 4 * i, fe_mul<F>(fe_load(scanned_totals + 4 * (blk - 1)), fe_load(out + 4 * i)));
}

// Reductions: Sum for ScalarShare / AuthenticatedScalarResult (share.rs:103-111, authenticated_scalar.rs:563-575: component-wise sums of
// shares and MACs) and Sum / Product for ScalarResult (scalar_result.rs:325-338; Iterator::sum over Scalars).  Field addition and
// multiplication are associative and commutative and results are canonical residues, so any tree gives the reference's value bit for
// bit.  Level 1: a grid-strided fold per thread and an LDS tree per workgroup -> one partial per workgroup; level 2: one workgroup over
// the partials.  blockIdx.y selects the column of a strided record (ScalarShare records: column 0 = shares, 1 = MACs).
#define RED_MAX_BLOCKS 1024
template <int F, int OP>   // OP_ADD or OP_MUL
__global__ void __launch_bounds__(TPB) k_reduce(size_t n, const u64* in, u32 stride, u32 col_off, u64* out, u32 out_stride) {
    __shared__ u64 sm[TPB * 4];
    const u64* src = in + (size_t)blockIdx.y * col_off;
    Fe acc = (OP == OP_MUL) ? fe_one<F>() : fe_zero<F>();
    for (size_t i = (size_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (size_t)gridDim.x * TPB) {
        const Fe v = fe_load(src + (size_t)stride * i);
        acc = (OP == OP_MUL) ? fe_mul<F>(acc, v) : fe_add<F>(acc, v);
    }
    fe_store(sm + 4 * threadIdx.x, acc);
    __syncthreads();
    for (int off = TPB / 2; off > 0; off >>= 1) {
        if (threadIdx.x < (unsigned)off) {
            const Fe a = fe_load(sm + 4 * threadIdx.x), b = fe_load(sm + 4 * (threadIdx.x + off));
            fe_store(sm + 4 * threadIdx.x, (OP == OP_MUL) ? fe_mul<F>(a, b) : fe_add<F>(a, b));
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) fe_store(out + (size_t)out_stride * blockIdx.x + (size_t)blockIdx.y * 4, fe_load(sm));
}

// ---------------------------------------------------------------------------------------------
// ScalarShare kernels (share.rs:72-133)
// ---------------------------------------------------------------------------------------------
// share.rs:85-131: add / sub / neg / mul(Scalar) are component-wise on (share, mac); add_public / sub_public (:74-82) add the
// public value to the share iff PARTY0 and mac_key * value to the MAC.
// "Flat" forms: a ScalarShare array is 2n consecutive field elements (share, mac, share,
// mac, ...), so one thread per ELEMENT (32 B, lanes 32 B apart) streams at the scalar kernels' 5.9 TB/s where one thread per
// 64-byte record reached 4.7-4.9 (tools/kernel_suite.py).  add / sub / neg are literally the scalar kernels over 2n elements.
template <int F>
__global__ void __launch_bounds__(TPB) k_share_mul_public_flat(size_t m, const u64* a, const u64* pub, u64* out) {      // m = 2n
    size_t j = (size_t)blockIdx.x * TPB + threadIdx.x;
    if (j >= m) return;
    fe_store(out + 4 * j, fe_mul<F>(fe_load(a + 4 * j), fe_load(pub + 4 * (j >> 1))));
}
template <int F, bool SUB>
__global__ void __launch_bounds__(TPB) k_share_addsub_public_flat(size_t m, int party, Fe key, const u64* a, const u64* pub, u64* out) {
    size_t j = (size_t)blockIdx.x * TPB + threadIdx.x;
    if (j >= m) return;
    Fe v = fe_load(a + 4 * j), r = fe_load(pub + 4 * (j >> 1));
    if (SUB) r = fe_neg<F>(r);
    if (j & 1) v = fe_add<F>(v, fe_mul<F>(key, r));          // mac element
    else if (party == 0) v = fe_add<F>(v, r);                // share element
    fe_store(out + 4 * j, v);
}
// column forms of the public-operand share ops (share.rs:74-82, :125-131): one thread per ScalarShare, share and MAC addressed through
// (pointer, stride) views -- the engine-native split columns, slices of them, or AoS records (stride 8, mac = share + 4)
template <int F, int OP>   // OP_ADD / OP_SUB: add_public / sub_public; OP_MUL: mul(Scalar)
__global__ void __launch_bounds__(TPB) k_share_public_v(size_t n, int party, Fe key, Col a_s, Col a_m, const u64* pub, ColOut o_s, ColOut o_m) {
    const size_t i = (size_t)blockIdx.x * TPB + threadIdx.x;
    if (i >= n) return;
    Fe r = fe_load(pub + 4 * i);
    const Fe s = fe_load(a_s.p + (size_t)a_s.stride * i), m = fe_load(a_m.p + (size_t)a_m.stride * i);
    if (OP == OP_MUL) {
        fe_store(o_s.p + (size_t)o_s.stride * i, fe_mul<F>(s, r));
        fe_store(o_m.p + (size_t)o_m.stride * i, fe_mul<F>(m, r));
    } else {
        if (OP == OP_SUB) r = fe_neg<F>(r);
        fe_store(o_s.p + (size_t)o_s.stride * i, party == 0 ? fe_add<F>(s, r) : s);
        fe_store(o_m.p + (size_t)o_m.stride * i, fe_add<F>(m, fe_mul<F>(key, r)));
    }
}
// layout converters between arkworks' AoS records and the engine-native split columns (field-independent copies)
__global__ void __launch_bounds__(TPB) k_share_split(size_t n, const u64* aos, u64* share_col, u64* mac_col) {
    size_t i = (size_t)blockIdx.x * TPB + threadIdx.x;
    if (i >= n) return;
    fe_store(share_col + 4 * i, fe_load(aos + 8 * i));
    fe_store(mac_col + 4 * i, fe_load(aos + 8 * i + 4));
}
__global__ void __launch_bounds__(TPB) k_share_join(size_t n, const u64* share_col, const u64* mac_col, u64* aos) {
    size_t i = (size_t)blockIdx.x * TPB + threadIdx.x;
    if (i >= n) return;
    fe_store(aos + 8 * i, fe_load(share_col + 4 * i));
    fe_store(aos + 8 * i + 4, fe_load(mac_col + 4 * i));
}
template <int F>
__global__ void __launch_bounds__(TPB) k_share_extract(size_t n, const u64* a, u64* out) {
    size_t i = (size_t)blockIdx.x * TPB + threadIdx.x;
    if (i >= n) return;
    fe_store(out + 4 * i, fe_load(a + 8 * i));
}

// ---------------------------------------------------------------------------------------------
// Beaver multiplication (authenticated_scalar.rs:848-879)
// ---------------------------------------------------------------------------------------------
// K1: d_i = x_i.share - a_i.share ; e_i = y_i.share - b_i.share ; out = d || e  (:863-868, :141-145).
// The MAC halves of d and e are dead in the reference (only `.share()` is sent), so they are not computed.
template <int F, int NT>   // NT bit0: x,y non-temporal ; bit1: a,b non-temporal ; bit2: d||e stores non-temporal
__global__ void __launch_bounds__(TPB) k_beaver_mask(size_t n, Col x, Col y, Col a, Col b, u64* out_d, u64* out_e, u64* dup_d, u64* dup_e, u32 xcd_blocks) {
    // xcd_blocks != 0 (a multiple of 8 = the grid size): workgroup b, which the dispatcher hands to XCD b % 8, takes the (b / 8)-th block of that
    // XCD's contiguous eighth of the batch instead of block b (experiment: ARKMPC_K1_XCD=1)
    const u32 blk = xcd_blocks ? (blockIdx.x & 7u) * (xcd_blocks >> 3) + (blockIdx.x >> 3) : blockIdx.x;
    size_t i = (size_t)blk * TPB + threadIdx.x;
    if (i >= n) return;
    Fe xs = (NT & 1) ? fe_load_nt(x.p + (size_t)x.stride * i) : fe_load(x.p + (size_t)x.stride * i);
    Fe as = (NT & 2) ? fe_load_nt(a.p + (size_t)a.stride * i) : fe_load(a.p + (size_t)a.stride * i);
    Fe ys = (NT & 1) ? fe_load_nt(y.p + (size_t)y.stride * i) : fe_load(y.p + (size_t)y.stride * i);
    Fe bs = (NT & 2) ? fe_load_nt(b.p + (size_t)b.stride * i) : fe_load(b.p + (size_t)b.stride * i);
    if (NT & 4) {
        fe_store_nt(out_d + 4 * i, fe_sub<F>(xs, as));
        fe_store_nt(out_e + 4 * i, fe_sub<F>(ys, bs));
    } else {
        fe_store(out_d + 4 * i, fe_sub<F>(xs, as));
        fe_store(out_e + 4 * i, fe_sub<F>(ys, bs));
    }
    if (dup_d) {            // the same payload a second time: the copy that is handed to the link (uniform branch)
        fe_store(dup_d + 4 * i, fe_sub<F>(xs, as));
        fe_store(dup_e + 4 * i, fe_sub<F>(ys, bs));
    }
}
// K2: open_batch combine gate (:161-171)
// (k_scalar_binop<F, OP_ADD> is used)

// K3 (+K2 when FUSED): [xy] = de + d[b] + e[a] + [c]  (:871-878 == :835-840)
//   share = d*b.share + e*a.share + c.share (+ d*e iff PARTY0)      (share.rs:74-77 add_public)
//   mac   = d*b.mac   + e*a.mac   + c.mac   + mac_key*(d*e)
// All sums are mod-p sums of canonical residues, so any association order is bit-identical to the
// reference's ((db + de) + (ea + c)).
template <int F, bool FUSED>
__global__ void __launch_bounds__(TPB) k_beaver_finish(size_t n, int party, Fe key, const u64* d0, const u64* e0, const u64* d1,
                                                       const u64* e1, Col a_s, Col a_m, Col b_s, Col b_m, Col c_s, Col c_m,
                                                       ColOut o_s, ColOut o_m) {
    size_t i = (size_t)blockIdx.x * TPB + threadIdx.x;
    if (i >= n) return;
    Fe d, e;
    if (FUSED) {  // d0, e0 = my d, e ; d1, e1 = the peer's
        d = fe_add<F>(fe_load(d0 + 4 * i), fe_load(d1 + 4 * i));
        e = fe_add<F>(fe_load(e0 + 4 * i), fe_load(e1 + 4 * i));
    } else {  // d0 = opened d, e0 = opened e
        d = fe_load(d0 + 4 * i);
        e = fe_load(e0 + 4 * i);
    }
    const Fe bs = fe_load(b_s.p + (size_t)b_s.stride * i), bm = fe_load(b_m.p + (size_t)b_m.stride * i);
    const Fe as = fe_load(a_s.p + (size_t)a_s.stride * i), am = fe_load(a_m.p + (size_t)a_m.stride * i);
    const Fe cs = fe_load(c_s.p + (size_t)c_s.stride * i), cm = fe_load(c_m.p + (size_t)c_m.stride * i);
    const Fe de = fe_mul<F>(d, e);
    Fe rs = fe_add<F>(fe_add<F>(fe_mul<F>(d, bs), fe_mul<F>(e, as)), cs);
    if (party == 0) rs = fe_add<F>(rs, de);
    Fe rm = fe_add<F>(fe_add<F>(fe_mul<F>(d, bm), fe_mul<F>(e, am)), fe_add<F>(cm, fe_mul<F>(key, de)));
    fe_store(o_s.p + (size_t)o_s.stride * i, rs);
    fe_store(o_m.p + (size_t)o_m.stride * i, rm);
}

// K2+K3, hand-scheduled: same semantics as k_beaver_finish<F, true>.  All 16 first-wave loads are issued before any
// arithmetic, the gate is regrouped to FIVE products under three Montgomery reductions (share = d (e [P0] + b.s) + e a.s + c.s,
// mac = d (key e + b.m) + e a.m + c.m: the same field elements as the reference's de + d[b] + e[a] + [c]), c is loaded over dead
// registers while the MAC rows run.  Byte offsets are 32-bit: the launcher chunks batches above 2^25 gates.
template <int F, int NT>
__global__ void __launch_bounds__(TPB) k_beaver_finish_asm(u32 n, u32 mask, Fe key, const u64* my_d, const u64* my_e, const u64* peer_d,
                                                           const u64* peer_e, const u64* a_s, const u64* a_m, const u64* b_s,
                                                           const u64* b_m, const u64* c_s, const u64* c_m, u64* out_s, u64* out_m,
                                                           u32 col_stride_bytes, u32 out_stride_bytes) {
    const u32 i = blockIdx.x * TPB + threadIdx.x;
    if (i >= n) return;
    beaver_finish_asm<F, NT>(i * 32u, i * col_stride_bytes, i * out_stride_bytes, my_d, my_e, peer_d, peer_e, a_s, a_m, b_s, b_m, c_s, c_m,
                         out_s, out_m, key, mask);
}

// AoS form of K2+K3 (arkworks ScalarShare records, 64 B = one cache line): the share and MAC halves of a record share a
// line, so the per-thread 16-byte loads of the body above touch every line with four instructions and the non-temporal hint
// backfires (measured: slower in every combination).  Here the workgroup copies its 256 a / b / c records into LDS with
// coalesced loads -- each line consumed by ONE instruction of four adjacent lanes, hint usable -- the body reads its record
// from LDS, leaves the result record in the a-region, and the workgroup streams the results out the same way.
template <int F>
__global__ void __launch_bounds__(TPB) k_beaver_finish_asm_aos(u32 n, u32 mask, Fe key, const u64* my_d, const u64* my_e, const u64* peer_d,
                                                               const u64* peer_e, const u64* a, const u64* b, const u64* c, u64* out) {
    typedef u32 v4u __attribute__((ext_vector_type(4)));
    __shared__ v4u lds[3 * TPB * 4];                         // 48 KiB: a | b | c, 256 records of 4 x 16 B each
    const u32 first = blockIdx.x * TPB, i = first + threadIdx.x;
    const u32 cnt = (n - first < TPB) ? n - first : TPB;     // records of this workgroup
    const v4u* a4 = reinterpret_cast<const v4u*>(a) + 4 * (size_t)first;
    const v4u* b4 = reinterpret_cast<const v4u*>(b) + 4 * (size_t)first;
    const v4u* c4 = reinterpret_cast<const v4u*>(c) + 4 * (size_t)first;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const u32 idx = threadIdx.x + k * TPB;
        if (idx < 4 * cnt) {
            lds[idx] = __builtin_nontemporal_load(a4 + idx);
            lds[4 * TPB + idx] = __builtin_nontemporal_load(b4 + idx);
            lds[8 * TPB + idx] = __builtin_nontemporal_load(c4 + idx);
        }
    }
    __syncthreads();
    const u32 lds_base = (u32)(size_t)(__attribute__((address_space(3))) char*)lds;
    if (i < n) beaver_finish_asm_lds<F>(i * 32u, lds_base + threadIdx.x * 64u, my_d, my_e, peer_d, peer_e, key, mask);
    __syncthreads();
    v4u* o4 = reinterpret_cast<v4u*>(out) + 4 * (size_t)first;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const u32 idx = threadIdx.x + k * TPB;
        if (idx < 4 * cnt) __builtin_nontemporal_store(lds[idx], o4 + idx);
    }
}

// Split-column K2+K3 with the result staged through LDS: the hand-scheduled body leaves each thread's share and MAC in two 8 KiB LDS
// columns and the workgroup writes them out lane-contiguously -- every non-temporal store instruction covers whole lines.  (The body's own
// stores are 16 bytes per lane, 32 bytes apart; as non-temporal stores they reach HBM as partial lines: +10 % write traffic measured.)
template <int F>
__global__ void __launch_bounds__(TPB) k_beaver_finish_asm_so(u32 n, u32 mask, Fe key, const u64* my_d, const u64* my_e, const u64* peer_d,
                                                              const u64* peer_e, const u64* a_s, const u64* a_m, const u64* b_s, const u64* b_m,
                                                              const u64* c_s, const u64* c_m, u64* out_s, u64* out_m) {
    typedef u32 v4u __attribute__((ext_vector_type(4)));
    __shared__ v4u lds[2 * TPB * 2];                         // 16 KiB: 256 shares (32 B each), then 256 MACs
    const u32 first = blockIdx.x * TPB, i = first + threadIdx.x;
    const u32 cnt = (n - first < TPB) ? n - first : TPB;
    const u32 lds_base = (u32)(size_t)(__attribute__((address_space(3))) char*)lds;
    if (i < n) beaver_finish_asm_so<F>(i * 32u, i * 32u, lds_base + threadIdx.x * 32u, my_d, my_e, peer_d, peer_e, a_s, a_m, b_s, b_m, c_s, c_m, key, mask);
    __syncthreads();
    v4u* os = reinterpret_cast<v4u*>(out_s) + 2 * (size_t)first;
    v4u* om = reinterpret_cast<v4u*>(out_m) + 2 * (size_t)first;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const u32 idx = threadIdx.x + k * TPB;
        if (idx < 2 * cnt) {
            __builtin_nontemporal_store(lds[idx], os + idx);
            __builtin_nontemporal_store(lds[2 * TPB + idx], om + idx);
        }
    }
}
// The same with WAVE-local staging: a wave writes out exactly the 128 chunks of each column its own 64 lanes produced (chunk 128 w + lane + 64 k),
// so the only ordering it needs is its own LDS writes (the body ends on s_waitcnt lgkmcnt(0)) -- no workgroup barrier, a wave retires without
// waiting for the other three (ARKMPC_K3_NT=4).
template <int F>
__global__ void __launch_bounds__(TPB) k_beaver_finish_asm_sw(u32 n, u32 mask, Fe key, const u64* my_d, const u64* my_e, const u64* peer_d,
                                                              const u64* peer_e, const u64* a_s, const u64* a_m, const u64* b_s, const u64* b_m,
                                                              const u64* c_s, const u64* c_m, u64* out_s, u64* out_m) {
    typedef u32 v4u __attribute__((ext_vector_type(4)));
    // 16 KiB are used; the allocation is 44 KiB so that a CU holds THREE workgroups (three waves per SIMD), not the four the 115 VGPRs would allow:
    // the kernel is bound by the memory system, not by latency -- measured with tools/k3_occupancy_probe.sh @77e688c: 4 / 3 / 2 / 1 waves per SIMD =
    // 57.8 / 57.0 / 59.4 / 82 us -- and the fourth wave only adds contention
    __shared__ v4u lds[2816];
    const u32 first = blockIdx.x * TPB, i = first + threadIdx.x;
    const u32 cnt = (n - first < TPB) ? n - first : TPB;
    const u32 lds_base = (u32)(size_t)(__attribute__((address_space(3))) char*)lds;
    if (i < n) beaver_finish_asm_so<F>(i * 32u, i * 32u, lds_base + threadIdx.x * 32u, my_d, my_e, peer_d, peer_e, a_s, a_m, b_s, b_m, c_s, c_m, key, mask);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    v4u* os = reinterpret_cast<v4u*>(out_s) + 2 * (size_t)first;
    v4u* om = reinterpret_cast<v4u*>(out_m) + 2 * (size_t)first;
    const u32 wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const u32 idx = 128u * wave + lane + 64u * k;
        if (idx < 2 * cnt) {
            __builtin_nontemporal_store(lds[idx], os + idx);
            __builtin_nontemporal_store(lds[2 * TPB + idx], om + idx);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// batch open + MAC check (authenticated_scalar.rs:278-354)
// ---------------------------------------------------------------------------------------------
// K4: chk_i = mac_key * opened_i - share_i.mac  (:299-311).  FUSED adds K2: opened_i = share_i.share + peer_i
template <int F, bool FUSED>
__global__ void __launch_bounds__(TPB) k_mac_check(size_t n, Fe key, const u64* opened_in, const u64* shares, const u64* peer,
                                                   u64* out_opened, u64* out_chk) {
    size_t i = (size_t)blockIdx.x * TPB + threadIdx.x;
    if (i >= n) return;
    Fe v;
    if (FUSED) {
        v = fe_add<F>(fe_load(shares + 8 * i), fe_load(peer + 4 * i));
        fe_store(out_opened + 4 * i, v);
    } else {
        v = fe_load(opened_in + 4 * i);
    }
    Fe mac = fe_load(shares + 8 * i + 4);
    fe_store(out_chk + 4 * i, fe_sub<F>(fe_mul<F>(key, v), mac));
}
// K2+K4 on share / MAC columns (engine-native split layout, or any strided view): the `.share()` payload a party sends IS its share
// column, so there is no extraction pass and the MAC half is read exactly once: 160 B per share here + 0 for the projection, against
// 96 + 160 on AoS records (which must be fetched whole twice)
template <int F>
__global__ void __launch_bounds__(TPB) k_mac_check_v(size_t n, Fe key, Col sh, Col mac, const u64* peer, u64* out_opened, u64* out_chk) {
    size_t i = (size_t)blockIdx.x * TPB + threadIdx.x;
    if (i >= n) return;
    const Fe v = fe_add<F>(fe_load(sh.p + (size_t)sh.stride * i), fe_load_nt(peer + 4 * i));
    fe_store(out_opened + 4 * i, v);
    fe_store(out_chk + 4 * i, fe_sub<F>(fe_mul<F>(key, v), fe_load_nt(mac.p + (size_t)mac.stride * i)));
}
// K5: all(mine_i + peer_i == 0)  (:218-219).  A wave with a failing element stores 1 into the context's verify flag, a word of
// host-coherent mapped memory: the success path writes nothing, so a call costs no memset and no read-back copy (round 1's
// memset + kernel + blocking D2H ran at 1.8 TB/s end to end).  Both inputs are streamed exactly once: non-temporal loads.
template <int F>
__global__ void __launch_bounds__(TPB) k_mac_verify(size_t n, const u64* mine, const u64* peer, int* host_flag, int* dev_gate) {
    size_t i = (size_t)blockIdx.x * TPB + threadIdx.x;
    bool bad = false;
    if (i < n) bad = !fe_is_zero(fe_add<F>(fe_load_nt(mine + 4 * i), fe_load_nt(peer + 4 * i)));
    // failure path: ONE store crosses to host memory per failed call -- a device-side gate word admits the first failing wave and
    // turns the rest away (an all-bad batch of 2^22 would otherwise issue 65536 PCIe writes: 7 ms instead of 50 us)
    if (__any(bad) && (threadIdx.x & 63) == 0 && __hip_atomic_load(dev_gate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0 &&
        atomicExch(dev_gate, 1) == 0)
        __hip_atomic_store(host_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// broadcast of one record of `w` 16-byte vectors to n slots (constant batches of a preprocessing source, fabric constants)
struct FillRecord { uint4 v[6]; };
__global__ void __launch_bounds__(TPB) k_fill_records(size_t n, u32 w, FillRecord rec, uint4* out) {
    const size_t t = (size_t)blockIdx.x * TPB + threadIdx.x;      // one 16-byte vector per thread: coalesced
    if (t >= n * w) return;
    const u32 k = (u32)(t % w);
    uint4 v = rec.v[0];
#pragma unroll
    for (int j = 1; j < 6; ++j) v = (k == (u32)j) ? rec.v[j] : v;
    out[t] = v;
}

// ---------------------------------------------------------------------------------------------
// dispatch helpers
// ---------------------------------------------------------------------------------------------
#define DISPATCH_FIELD(ctx, EXPR_F)                                                             \
    switch ((ctx)->field_id) {                                                                  \
        case 0: { constexpr int F = 0; EXPR_F; } break;                                         \
        case 1: { constexpr int F = 1; EXPR_F; } break;                                         \
        case 2: { constexpr int F = 2; EXPR_F; } break;                                         \
        case 3: { constexpr int F = 3; EXPR_F; } break;                                         \
        case 4: { constexpr int F = 4; EXPR_F; } break;                                         \
        default: return ark_bad(ctx, "bad field id");                                           \
    }

#define ENTER(ctx)                                                                              \
    if (!(ctx)) return ARKMPC_ERR_BAD_ARG;                                                      \
    CtxGuard guard__(ctx);                                                                      \
    if (guard__.rc) return guard__.rc;

static inline bool party_ok(int p) { return p == 0 || p == 1; }

static int k1_nt_mode() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("ARKMPC_K1_NT"); v = e ? atoi(e) & 7 : 1; }   // default 1: x,y streamed once -> non-temporal (measured best; a,b and d||e are re-read by K3)
    return v;
}
template <int F>
static void launch_mask(arkmpc_ctx* ctx, size_t n, Col x, Col y, Col a, Col b, u64* out, u64* out_e = nullptr, u64* dup = nullptr) {
    if (!out_e) out_e = out + 4 * n;                     // d||e in one buffer unless the caller places e itself
    u64* dup_e = dup ? dup + 4 * n : nullptr;
    dim3 g(blocks_for(n, TPB)), t(TPB);
    const bool split = x.stride == 4 && y.stride == 4 && a.stride == 4 && b.stride == 4;
    static const int aos_mode = getenv("ARKMPC_K1_NT_AOS") ? atoi(getenv("ARKMPC_K1_NT_AOS")) & 7 : 0;
    static const bool xcd_map = getenv("ARKMPC_K1_XCD") && getenv("ARKMPC_K1_XCD")[0] == '1';
    const u32 xcd = (xcd_map && g.x % 8 == 0 && n % TPB == 0) ? g.x : 0u;
    // unused dynamic LDS caps the workgroups per CU (160 KB per CU): K1 is bound by the memory system and runs best with few waves in flight
    // (tools/k1_occupancy_probe.sh @77e688c, 2^20 gates: 8+ / 4 / 3 / 2 / 1 workgroups per CU = 36.5 / 36.0 / 35.9 / 35.1 / 36.0 us): two workgroups per CU for large batches
    static const int k1_lds_env = getenv("ARKMPC_K1_LDS") ? atoi(getenv("ARKMPC_K1_LDS")) : -1;
    // the cap is only applied where a workgroup may own that much LDS (gfx950: 160 KB); elsewhere the launch would fail, so it is dropped
    if (!ctx->lds_per_wg) { int v = 0; ctx->lds_per_wg = hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, ctx->device) == hipSuccess && v > 0 ? (unsigned)v : 65536u; }
    const unsigned lds_per_wg = ctx->lds_per_wg;
    unsigned k1_lds = k1_lds_env >= 0 ? (unsigned)k1_lds_env : (n >= ((size_t)1 << 16) ? 80000u : 0u);
    if (k1_lds > lds_per_wg) k1_lds = 0;
    switch (split ? k1_nt_mode() : aos_mode) {
        case 1: launch_k_lds(ctx, k1_lds, k_beaver_mask<F, 1>, g, t, n, x, y, a, b, out, out_e, dup, dup_e, xcd); break;
        case 2: launch_k_lds(ctx, k1_lds, k_beaver_mask<F, 2>, g, t, n, x, y, a, b, out, out_e, dup, dup_e, xcd); break;
        case 3: launch_k_lds(ctx, k1_lds, k_beaver_mask<F, 3>, g, t, n, x, y, a, b, out, out_e, dup, dup_e, xcd); break;
        case 4: launch_k_lds(ctx, k1_lds, k_beaver_mask<F, 4>, g, t, n, x, y, a, b, out, out_e, dup, dup_e, xcd); break;
        case 5: launch_k_lds(ctx, k1_lds, k_beaver_mask<F, 5>, g, t, n, x, y, a, b, out, out_e, dup, dup_e, xcd); break;
        case 7: launch_k_lds(ctx, k1_lds, k_beaver_mask<F, 7>, g, t, n, x, y, a, b, out, out_e, dup, dup_e, xcd); break;
        default: launch_k_lds(ctx, k1_lds, k_beaver_mask<F, 0>, g, t, n, x, y, a, b, out, out_e, dup, dup_e, xcd); break;
    }
}

static bool use_asm_path() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("ARKMPC_NO_ASM"); v = (e && e[0] == '1') ? 0 : 1; }
    return v == 1;
}

// Launch K2+K3 over n gates (device pointers).  Uses the hand-scheduled body when the field has one and the three
// triple columns share a stride; otherwise the C++ kernel.
template <int F>
static void launch_finish_fused(arkmpc_ctx* ctx, size_t n, int party, const Fe& k, const u64* my_de, const u64* peer_de, Col a_s, Col a_m,
                                Col b_s, Col b_m, Col c_s, Col c_m, ColOut o_s, ColOut o_m, const u64* my_e = nullptr, const u64* peer_e = nullptr) {
    const u64* my_d = my_de; const u64* peer_d = peer_de;
    if (!my_e) my_e = my_de + 4 * n;                     // d||e in one buffer unless the caller addresses e itself (a range of a larger batch)
    if (!peer_e) peer_e = peer_de + 4 * n;
    if constexpr (HasAsmFinish<F>::value) {
        // the hand-scheduled body addresses with 32-bit byte offsets inside a chunk of 2^25 gates: (2^25 - 1) * stride * 8 fits only
        // for strides up to 16 u64 (AoS = 8, split = 4); wider views take the 64-bit C++ kernel below
        if (use_asm_path() && a_s.stride == b_s.stride && a_s.stride == c_s.stride && a_s.stride <= 16 && o_s.stride <= 16) {
            const size_t CH = (size_t)1 << 25;
            const u32 mask = party == 0 ? 0xffffffffu : 0u;
            for (size_t lo = 0; lo < n; lo += CH) {
                const size_t cnt = (n - lo < CH) ? (n - lo) : CH;
                const size_t cs = (size_t)a_s.stride * lo, os = (size_t)o_s.stride * lo;
                // split-column layout (stride 4): once-streamed data carries non-temporal hints; AoS: none (halves share lines)
                // ARKMPC_K3_NT (A/B switch, split columns; measured at 2^20 gates, tools/ab_k3_nt.sh @77e688c): 3 (default) = non-temporal hints on the
                // once-streamed loads, result staged through LDS and written with whole-line non-temporal stores: 58.4 us; 1 = the same hints with the
                // body's own 16-byte stores: 61.1 us (partial-line write amplification); 2 = hints on the loads only: 66 us (plain stores keep the
                // results in the cache K3's re-reads want); 0 = no hints: 74 us
                static const int k3_nt = getenv("ARKMPC_K3_NT") ? atoi(getenv("ARKMPC_K3_NT")) : 4;     // 4: LDS-staged whole-line stores, per wave (measured best; 3: per workgroup; 1 / 2: the body's own stores)
                static const bool aos_lds = !(getenv("ARKMPC_K3_AOS_LDS") && getenv("ARKMPC_K3_AOS_LDS")[0] == '0');
                const bool aos_records = a_s.stride == 8 && o_s.stride == 8 && a_m.p == a_s.p + 4 && b_m.p == b_s.p + 4 && c_m.p == c_s.p + 4 &&
                                         o_m.p == o_s.p + 4;
                if (aos_records && aos_lds) {
                    launch_k(ctx, k_beaver_finish_asm_aos<F>, dim3(blocks_for(cnt, TPB)), dim3(TPB), (u32)cnt, mask, k, my_d + 4 * lo,
                             my_e + 4 * lo, peer_d + 4 * lo, peer_e + 4 * lo, a_s.p + cs, b_s.p + cs, c_s.p + cs, o_s.p + os);
                    continue;
                }
                const int nt = (a_s.stride == 4 && o_s.stride == 4) ? k3_nt : 0;
#define ARK_K3_LAUNCH(NT)                                                                                                                           \
    launch_k(ctx, k_beaver_finish_asm<F, NT>, dim3(blocks_for(cnt, TPB)), dim3(TPB), (u32)cnt, mask, k, my_d + 4 * lo, my_e + 4 * lo, peer_d + 4 * lo,   \
             peer_e + 4 * lo, a_s.p + cs, a_m.p + cs, b_s.p + cs, b_m.p + cs, c_s.p + cs, c_m.p + cs, o_s.p + os, o_m.p + os, a_s.stride * 8u,          \
             o_s.stride * 8u)
                if (nt == 3)            // result staged through LDS, whole-line stores (split columns only)
                    launch_k(ctx, k_beaver_finish_asm_so<F>, dim3(blocks_for(cnt, TPB)), dim3(TPB), (u32)cnt, mask, k, my_d + 4 * lo, my_e + 4 * lo, peer_d + 4 * lo,
                             peer_e + 4 * lo, a_s.p + cs, a_m.p + cs, b_s.p + cs, b_m.p + cs, c_s.p + cs, c_m.p + cs, o_s.p + os, o_m.p + os);
                else if (nt == 4)       // the same with wave-local staging (no workgroup barrier)
                    launch_k(ctx, k_beaver_finish_asm_sw<F>, dim3(blocks_for(cnt, TPB)), dim3(TPB), (u32)cnt, mask, k, my_d + 4 * lo, my_e + 4 * lo, peer_d + 4 * lo,
                             peer_e + 4 * lo, a_s.p + cs, a_m.p + cs, b_s.p + cs, b_m.p + cs, c_s.p + cs, c_m.p + cs, o_s.p + os, o_m.p + os);
                else if (nt == 1) ARK_K3_LAUNCH(1);
                else if (nt == 2) ARK_K3_LAUNCH(2);
                else ARK_K3_LAUNCH(0);
#undef ARK_K3_LAUNCH
            }
            return;
        }
    }
    launch_k(ctx, k_beaver_finish<F, true>, dim3(blocks_for(n, TPB)), dim3(TPB), n, party, k, my_d, my_e, peer_d, peer_e, a_s, a_m,
             b_s, b_m, c_s, c_m, o_s, o_m);
}

template <int F>
static void scan_level(arkmpc_ctx* ctx, size_t n, const u64* in, u64* out, u64* ws) {
    const size_t nblocks = (n + SCAN_BLOCK - 1) / SCAN_BLOCK;
    u64* totals = ws;                                   // nblocks elements, then the deeper levels' workspace
    hipLaunchKernelGGL((k_scan_block<F>), dim3((unsigned)nblocks), dim3(TPB), 0, ctx->stream, n, in, out, nblocks > 1 ? totals : (u64*)nullptr);
    if (nblocks > 1) {
        scan_level<F>(ctx, nblocks, totals, totals, totals + 4 * nblocks);      // in-place scan of the block totals
        hipLaunchKernelGGL((k_scan_apply<F>), dim3(blocks_for(n, TPB)), dim3(TPB), 0, ctx->stream, n, out, totals);
    }
}

// ---------------------------------------------------------------------------------------------
// C ABI: context
// ---------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------------
// Device memory pool behind arkmpc_malloc / arkmpc_free.  hipFree synchronises the whole device and unmaps the range
// (about a millisecond for the 32-64 MiB batches of this path); a host-side protocol step allocates and drops a dozen of
// them.  Freed blocks are kept per size class (power of two below 1 MiB, 2 MiB granules above) and handed out again.
// arkmpc_free is STREAM-ORDERED (hipFreeAsync semantics): it records an event on the context's stream and the block
// becomes reusable once that event has completed -- no host-side wait.  A buffer must therefore be freed through the
// context whose stream last used it (the host mirror rebinds a batch received from the peer party to the receiver's
// engine).  ARKMPC_NO_POOL=1 restores plain hipMalloc / hipFree; ARKMPC_POOL_MAX_MB caps the cached bytes (default 16 GiB
// of the 288 GB).
// ---------------------------------------------------------------------------------------------
#include <map>
#include <unordered_map>
namespace {
struct PendingBlock { void* ptr; size_t cls; hipEvent_t ev; };
struct DevicePool {
    std::mutex mu;
    std::map<size_t, std::vector<void*>> free_;
    std::vector<PendingBlock> pending_;      // freed, waiting for the freeing stream to pass the event
    std::vector<hipEvent_t> events_;         // recycled events
    std::unordered_map<void*, size_t> live_;
    size_t cached = 0;                       // bytes in free_ + pending_
    int refs = 0;
};
DevicePool g_pool[16];
bool pool_enabled() {
    static const bool on = !(getenv("ARKMPC_NO_POOL") && getenv("ARKMPC_NO_POOL")[0] == '1');
    return on;
}
size_t pool_cap() {
    static const size_t cap = getenv("ARKMPC_POOL_MAX_MB") ? (size_t)atoll(getenv("ARKMPC_POOL_MAX_MB")) << 20 : (size_t)16 << 30;
    return cap;
}
size_t ctx_cache_cap() {      // bytes a context keeps in its own event-free cache before spilling to the device pool
    static const size_t cap = getenv("ARKMPC_CTX_CACHE_MB") ? (size_t)atoll(getenv("ARKMPC_CTX_CACHE_MB")) << 20 : (size_t)2 << 30;
    return cap;
}
size_t pool_class(size_t bytes) {
    if (bytes < 256) bytes = 256;
    if (bytes >= ((size_t)1 << 20)) return (bytes + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
    size_t c = 256;
    while (c < bytes) c <<= 1;
    return c;
}
// move every pending block whose event has completed to the free lists (caller holds p.mu)
void pool_reclaim(DevicePool& p, bool wait) {
    size_t k = 0;
    for (size_t i = 0; i < p.pending_.size(); ++i) {
        PendingBlock& b = p.pending_[i];
        const bool done = wait ? (hipEventSynchronize(b.ev) == hipSuccess) : (hipEventQuery(b.ev) == hipSuccess);
        if (done) { p.free_[b.cls].push_back(b.ptr); p.events_.push_back(b.ev); }
        else p.pending_[k++] = b;
    }
    p.pending_.resize(k);
    (void)hipGetLastError();     // hipEventQuery reports "not ready" through the error state
}
void pool_release_all(DevicePool& p) {   // caller holds p.mu; blocks everything pending first
    pool_reclaim(p, true);
    for (auto& kv : p.free_) for (void* q : kv.second) { (void)hipFree(q); p.cached -= kv.first; }     // (blocks in context caches stay counted)
    p.free_.clear();
    for (hipEvent_t e : p.events_) (void)hipEventDestroy(e);
    p.events_.clear();
}
}  // namespace

extern "C" {

const char* arkmpc_version(void) { return "arkmpc-hip 0.1 (gfx950)"; }

int arkmpc_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int arkmpc_ctx_create(int field_id, int device, arkmpc_ctx** out_ctx) {
    if (!out_ctx) return ARKMPC_ERR_BAD_ARG;
    *out_ctx = nullptr;
    if (field_id < 0 || field_id >= F_NFIELDS) return ARKMPC_ERR_BAD_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return ARKMPC_ERR_NO_DEVICE;
    if (device < 0 || device >= ndev) return ARKMPC_ERR_BAD_ARG;
    if (hipSetDevice(device) != hipSuccess) return ARKMPC_ERR_HIP;
    arkmpc_ctx* c = new arkmpc_ctx();
    c->field_id = field_id;
    c->device = device;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return ARKMPC_ERR_HIP; }
    c->own_stream = true;
    if (hipMalloc((void**)&c->d_flag, 64) != hipSuccess || hipHostMalloc((void**)&c->h_flag, 64) != hipSuccess ||
        hipHostMalloc((void**)&c->h_vflag, 64, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
        hipHostGetDevicePointer((void**)&c->d_vflag, c->h_vflag, 0) != hipSuccess) {
        if (c->d_flag) (void)hipFree(c->d_flag);
        if (c->h_flag) (void)hipHostFree(c->h_flag);
        if (c->h_vflag) (void)hipHostFree(c->h_vflag);
        delete c;
        return ARKMPC_ERR_HIP;
    }
    *c->h_vflag = 0;
    c->d_vgate = c->d_flag + 8;                         // second word group of the 64-byte device flag block
    if (hipMemset(c->d_flag, 0, 64) != hipSuccess) { (void)hipFree(c->d_flag); (void)hipHostFree(c->h_flag); (void)hipHostFree(c->h_vflag); delete c; return ARKMPC_ERR_HIP; }
    if (device < 16) { std::lock_guard<std::mutex> lk(g_pool[device].mu); g_pool[device].refs++; }
    if (device < 32) devices_in_use().fetch_or(1u << device, std::memory_order_relaxed);
    *out_ctx = c;
    return ARKMPC_OK;
}

int arkmpc_ctx_destroy(arkmpc_ctx* ctx) {
    if (!ctx) return ARKMPC_ERR_BAD_ARG;
    if (ctx->device < 16) {          // the last context of a device returns the cached blocks
        DevicePool& p = g_pool[ctx->device];
        std::lock_guard<std::mutex> lk(p.mu);
        (void)hipSetDevice(ctx->device);
        (void)hipStreamSynchronize(ctx->stream);         // the context's own cache joins the pool: its stream has drained, no event needed
        for (auto& kv : ctx->cache) for (void* q : kv.second) p.free_[kv.first].push_back(q);
        ctx->cache.clear(); ctx->cache_bytes = 0;
        if (--p.refs == 0) { (void)hipSetDevice(ctx->device); (void)hipDeviceSynchronize(); pool_release_all(p); }
    }
    {
        std::lock_guard<std::mutex> lk(ctx->mu);
        (void)hipSetDevice(ctx->device);
        (void)hipStreamSynchronize(ctx->stream);
        if (ctx->arena) (void)hipFree(ctx->arena);
        if (ctx->d_flag) (void)hipFree(ctx->d_flag);
        if (ctx->h_flag) (void)hipHostFree(ctx->h_flag);
        if (ctx->h_vflag) (void)hipHostFree(ctx->h_vflag);
        if (ctx->h_small) (void)hipHostFree(ctx->h_small);
        for (int i = 0; i < 2; ++i) {
            if (ctx->h_pin[i]) (void)hipHostFree(ctx->h_pin[i]);
            if (ctx->ev[i]) (void)hipEventDestroy(ctx->ev[i]);
        }
        for (auto& e : ctx->tev) if (e) (void)hipEventDestroy(e);
        for (auto& e : ctx->link_ev) (void)hipEventDestroy(e);
        if (ctx->up) { (void)hipStreamSynchronize(ctx->up); (void)hipStreamDestroy(ctx->up); }
        if (ctx->down) { (void)hipStreamSynchronize(ctx->down); (void)hipStreamDestroy(ctx->down); }
        if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    }
    delete ctx;
    return ARKMPC_OK;
}

int arkmpc_ctx_set_stream(arkmpc_ctx* ctx, void* hip_stream) {
    ENTER(ctx);
    ARK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->own_stream && ctx->stream) { ARK_HIP(ctx, hipStreamDestroy(ctx->stream)); }
    ctx->stream = (hipStream_t)hip_stream;
    ctx->own_stream = false;
    return ARKMPC_OK;
}

int arkmpc_ctx_set_host_buffers(arkmpc_ctx* ctx, int enabled) {
    ENTER(ctx);
    ctx->host_buffers = enabled != 0;
    return ARKMPC_OK;
}

int arkmpc_sync(arkmpc_ctx* ctx) {
    ENTER(ctx);
    ARK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ARKMPC_OK;
}

int arkmpc_kernel_timer_arm(arkmpc_ctx* ctx, int slot) {
    ENTER(ctx);
    if (slot < 0 || slot >= arkmpc_ctx::kTimerSlots) return ark_bad(ctx, "timer slot out of range");
    for (int i = 0; i < 2; ++i)
        if (!ctx->tev[2 * slot + i]) ARK_HIP(ctx, hipEventCreate(&ctx->tev[2 * slot + i]));
    ctx->timer_slot = slot;
    return ARKMPC_OK;
}
int arkmpc_kernel_timer_ms(arkmpc_ctx* ctx, int slot, float* out_ms) {
    ENTER(ctx);
    if (slot < 0 || slot >= arkmpc_ctx::kTimerSlots || !out_ms || !ctx->tev[2 * slot]) return ark_bad(ctx, "timer slot not armed");
    ARK_HIP(ctx, hipEventSynchronize(ctx->tev[2 * slot + 1]));
    ARK_HIP(ctx, hipEventElapsedTime(out_ms, ctx->tev[2 * slot], ctx->tev[2 * slot + 1]));
    return ARKMPC_OK;
}

// The message is copied under the error-text lock into a per-thread buffer: a concurrent call on the same context may be
// rewriting ctx->err (std::string reallocation) while this thread reads it.  (Not the context lock: asking for the last
// error must not wait behind a long-running call of another thread.)
const char* arkmpc_last_error(arkmpc_ctx* ctx) {
    if (!ctx) return "null context";
    static thread_local std::string snapshot;
    { std::lock_guard<std::mutex> lk(ctx->err_mu); snapshot = ctx->err; }
    return snapshot.c_str();
}

// body of arkmpc_malloc; the caller holds the context lock (ENTER)
static int ark_malloc_locked(arkmpc_ctx* ctx, size_t bytes, void** out_dptr) {
    if (!out_dptr) return ark_bad(ctx, "null out pointer");
    if (!pool_enabled() || ctx->device >= 16) {
        ARK_HIP(ctx, hipMalloc(out_dptr, bytes ? bytes : 16));
        return ARKMPC_OK;
    }
    DevicePool& p = g_pool[ctx->device];
    const size_t cls = pool_class(bytes);
    std::lock_guard<std::mutex> lk(p.mu);
    {   // the context's own cache: no event, no query -- stream order is the guarantee
        auto ci = ctx->cache.find(cls);
        if (ci != ctx->cache.end() && !ci->second.empty()) {
            *out_dptr = ci->second.back();
            ci->second.pop_back();
            ctx->cache_bytes -= cls;
            p.cached -= cls;
            p.live_[*out_dptr] = cls;
            return ARKMPC_OK;
        }
    }
    auto it = p.free_.find(cls);
    if (it == p.free_.end() || it->second.empty()) { pool_reclaim(p, false); it = p.free_.find(cls); }
    if (it != p.free_.end() && !it->second.empty()) {
        *out_dptr = it->second.back();
        it->second.pop_back();
        p.cached -= cls;
    } else {
        hipError_t e = hipMalloc(out_dptr, cls);
        if (e != hipSuccess && p.cached) {       // out of memory with blocks cached: give them back and retry
            (void)hipGetLastError();
            (void)hipStreamSynchronize(ctx->stream);
            for (auto& kv : ctx->cache) for (void* q : kv.second) { (void)hipFree(q); p.cached -= kv.first; }
            ctx->cache.clear(); ctx->cache_bytes = 0;
            pool_release_all(p);
            e = hipMalloc(out_dptr, cls);
        }
        if (e != hipSuccess) { ark_set_err(ctx, std::string("hipMalloc: ") + hipGetErrorString(e)); return ARKMPC_ERR_HIP; }
    }
    p.live_[*out_dptr] = cls;
    return ARKMPC_OK;
}
int arkmpc_malloc(arkmpc_ctx* ctx, size_t bytes, void** out_dptr) {
    ENTER(ctx);
    return ark_malloc_locked(ctx, bytes, out_dptr);
}
// body of arkmpc_free; the caller holds the context lock
static int ark_free_locked(arkmpc_ctx* ctx, void* dptr) {
    if (!dptr) return ARKMPC_OK;
    if (pool_enabled() && ctx->device < 16) {
        DevicePool& p = g_pool[ctx->device];
        std::lock_guard<std::mutex> lk(p.mu);
        auto it = p.live_.find(dptr);
        if (it != p.live_.end()) {
            const size_t cls = it->second;
            p.live_.erase(it);
            if (ctx->cache_bytes + cls <= ctx_cache_cap() && p.cached + cls <= pool_cap()) {     // keep it for this context: no event needed
                ctx->cache[cls].push_back(dptr);
                ctx->cache_bytes += cls;
                p.cached += cls;
                return ARKMPC_OK;
            }
            if (p.cached + cls > pool_cap()) {           // over the cap: a real free (waits for the device)
                ARK_HIP(ctx, hipStreamSynchronize(ctx->stream));
                ARK_HIP(ctx, hipFree(dptr));
                return ARKMPC_OK;
            }
            hipEvent_t ev;
            if (!p.events_.empty()) { ev = p.events_.back(); p.events_.pop_back(); }
            else ARK_HIP(ctx, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
            ARK_HIP(ctx, hipEventRecord(ev, ctx->stream));
            p.pending_.push_back({dptr, cls, ev});
            p.cached += cls;
            return ARKMPC_OK;
        }
    }
    ARK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ARK_HIP(ctx, hipFree(dptr));
    return ARKMPC_OK;
}
int arkmpc_free(arkmpc_ctx* ctx, void* dptr) {
    ENTER(ctx);
    return ark_free_locked(ctx, dptr);
}
int arkmpc_memcpy_h2d(arkmpc_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes) {
    ENTER(ctx);
    ARK_HIP(ctx, hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, ctx->stream));
    ARK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ARKMPC_OK;
}
int arkmpc_memcpy_d2h(arkmpc_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes) {
    ENTER(ctx);
    ARK_HIP(ctx, hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
    ARK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ARKMPC_OK;
}

// ---------------------------------------------------------------------------------------------
// C ABI: Scalar vectors
// ---------------------------------------------------------------------------------------------
static int scalar_binop(arkmpc_ctx* ctx, int op, size_t n, const uint64_t* a, const uint64_t* b, uint64_t* out) {
    ENTER(ctx);
    Stage st(ctx);
    int ia = st.declare_in(a, n * 32), ib = st.declare_in(b, n * 32), io = st.declare_out(out, n * 32);
    st.elementwise(n);
    if (st.commit()) return st.rc;
    size_t lo, cnt;
    while (st.next_chunk(&lo, &cnt)) {
        const u64 *da = st.in<u64>(ia) + 4 * lo, *db = st.in<u64>(ib) + 4 * lo;
        u64* dout = st.out<u64>(io) + 4 * lo;
        dim3 g(blocks_for(cnt, TPB)), t(TPB);
        DISPATCH_FIELD(ctx, {
            if (op == OP_ADD) hipLaunchKernelGGL((k_scalar_binop<F, OP_ADD>), g, t, 0, ctx->stream, cnt, da, db, dout);
            if (op == OP_SUB) hipLaunchKernelGGL((k_scalar_binop<F, OP_SUB>), g, t, 0, ctx->stream, cnt, da, db, dout);
            if (op == OP_MUL) hipLaunchKernelGGL((k_scalar_binop<F, OP_MUL>), g, t, 0, ctx->stream, cnt, da, db, dout);
        });
    }
    return st.finish();
}
int arkmpc_scalar_add(arkmpc_ctx* ctx, size_t n, const uint64_t* a, const uint64_t* b, uint64_t* out) { return scalar_binop(ctx, OP_ADD, n, a, b, out); }
int arkmpc_scalar_sub(arkmpc_ctx* ctx, size_t n, const uint64_t* a, const uint64_t* b, uint64_t* out) { return scalar_binop(ctx, OP_SUB, n, a, b, out); }
int arkmpc_scalar_mul(arkmpc_ctx* ctx, size_t n, const uint64_t* a, const uint64_t* b, uint64_t* out) { return scalar_binop(ctx, OP_MUL, n, a, b, out); }
int arkmpc_open_combine(arkmpc_ctx* ctx, size_t n, const uint64_t* mine, const uint64_t* peer, uint64_t* out) { return scalar_binop(ctx, OP_ADD, n, mine, peer, out); }

static int scalar_unop(arkmpc_ctx* ctx, int which, size_t n, const uint64_t* a, void* out, size_t out_elem_bytes) {
    ENTER(ctx);
    Stage st(ctx);
    int ia = st.declare_in(a, n * 32), io = st.declare_out(out, n * out_elem_bytes);
    st.elementwise(n);
    if (st.commit()) return st.rc;
    size_t lo, cnt;
    while (st.next_chunk(&lo, &cnt)) {
        const u64* da = st.in<u64>(ia) + 4 * lo;
        unsigned char* dout = st.out<unsigned char>(io) + out_elem_bytes * lo;
        dim3 g(blocks_for(cnt, TPB)), t(TPB);
        DISPATCH_FIELD(ctx, {
            if (which == 0) hipLaunchKernelGGL((k_scalar_neg<F>), g, t, 0, ctx->stream, cnt, da, (u64*)dout);
            if (which == 1) hipLaunchKernelGGL((k_from_canonical<F>), g, t, 0, ctx->stream, cnt, da, (u64*)dout);
            if (which == 2) hipLaunchKernelGGL((k_to_canonical<F>), g, t, 0, ctx->stream, cnt, da, (u64*)dout);
            if (which == 3) hipLaunchKernelGGL((k_to_bytes_be<F>), g, t, 0, ctx->stream, cnt, da, dout);
        });
    }
    return st.finish();
}
int arkmpc_scalar_prefix_product(arkmpc_ctx* ctx, size_t n, const uint64_t* a, uint64_t* out) {
    ENTER(ctx);
    Stage st(ctx);
    int ia = st.declare_in(a, n * 32), io = st.declare_out(out, n * 32);
    size_t ws = 0;
    for (size_t m = (n + SCAN_BLOCK - 1) / SCAN_BLOCK; m > 1; m = (m + SCAN_BLOCK - 1) / SCAN_BLOCK) ws += m * 32;
    ws += 64;
    int iw = st.declare_scratch(ws + ((n + SCAN_BLOCK - 1) / SCAN_BLOCK) * 32);
    if (st.commit()) return st.rc;
    if (n) DISPATCH_FIELD(ctx, scan_level<F>(ctx, n, st.in<u64>(ia), st.out<u64>(io), st.scratch<u64>(iw)));
    return st.finish();
}
int arkmpc_memcpy_d2d(arkmpc_ctx* ctx, void* dst_dev, const void* src_dev, size_t bytes) {
    ENTER(ctx);
    ARK_HIP(ctx, hipMemcpyAsync(dst_dev, src_dev, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    return ARKMPC_OK;
}
int arkmpc_scalar_batch_inverse(arkmpc_ctx* ctx, size_t n, const uint64_t* a, uint64_t* out) {
    ENTER(ctx);
    Stage st(ctx);
    int ia = st.declare_in(a, n * 32), io = st.declare_out(out, n * 32), ip = st.declare_scratch(n * 32 + 32);
    if (st.commit()) return st.rc;
    if (n) {
        size_t k = n >> 15;                                   // elements per thread: 8 .. 64, about 2^15 .. 2^16 threads for large batches
        const u32 K = (u32)(k < 8 ? 8 : (k > 64 ? 64 : k));
        const size_t threads = (n + K - 1) / K;
        DISPATCH_FIELD(ctx, hipLaunchKernelGGL((k_batch_inverse<F>), dim3(blocks_for(threads, TPB)), dim3(TPB), 0, ctx->stream, n, K, threads, st.in<u64>(ia),
                                               st.scratch<u64>(ip), st.out<u64>(io)));
    }
    return st.finish();
}
// sum / product over n elements of `cols` interleaved columns (element i of column c at in + stride * i + 4 c); out = cols elements
static int scalar_reduce(arkmpc_ctx* ctx, int op, size_t n, const uint64_t* in, u32 cols, uint64_t* out) {
    ENTER(ctx);
    Stage st(ctx);
    const u32 stride = 4 * cols;
    int ia = st.declare_in(in, n * stride * 8), io = st.declare_out(out, (size_t)cols * 32);
    const unsigned nb = (unsigned)((n + TPB - 1) / TPB < RED_MAX_BLOCKS ? (n + TPB - 1) / TPB : RED_MAX_BLOCKS);
    int iw = st.declare_scratch((size_t)(nb ? nb : 1) * cols * 32);
    if (st.commit()) return st.rc;
    DISPATCH_FIELD(ctx, {
        u64* part = st.scratch<u64>(iw);
        if (nb <= 1) {                                      // one workgroup folds everything (n = 0: the empty sum / product)
            if (op == OP_ADD) hipLaunchKernelGGL((k_reduce<F, OP_ADD>), dim3(1, cols), dim3(TPB), 0, ctx->stream, n, st.in<u64>(ia), stride, 4u, st.out<u64>(io), 0u);
            else hipLaunchKernelGGL((k_reduce<F, OP_MUL>), dim3(1, cols), dim3(TPB), 0, ctx->stream, n, st.in<u64>(ia), stride, 4u, st.out<u64>(io), 0u);
        } else {
            if (op == OP_ADD) {
                hipLaunchKernelGGL((k_reduce<F, OP_ADD>), dim3(nb, cols), dim3(TPB), 0, ctx->stream, n, st.in<u64>(ia), stride, 4u, part, stride);
                hipLaunchKernelGGL((k_reduce<F, OP_ADD>), dim3(1, cols), dim3(TPB), 0, ctx->stream, (size_t)nb, (const u64*)part, stride, 4u, st.out<u64>(io), 0u);
            } else {
                hipLaunchKernelGGL((k_reduce<F, OP_MUL>), dim3(nb, cols), dim3(TPB), 0, ctx->stream, n, st.in<u64>(ia), stride, 4u, part, stride);
                hipLaunchKernelGGL((k_reduce<F, OP_MUL>), dim3(1, cols), dim3(TPB), 0, ctx->stream, (size_t)nb, (const u64*)part, stride, 4u, st.out<u64>(io), 0u);
            }
        }
    });
    return st.finish();
}
int arkmpc_scalar_sum(arkmpc_ctx* ctx, size_t n, const uint64_t* a, uint64_t* out) { return scalar_reduce(ctx, OP_ADD, n, a, 1, out); }
int arkmpc_scalar_product(arkmpc_ctx* ctx, size_t n, const uint64_t* a, uint64_t* out) { return scalar_reduce(ctx, OP_MUL, n, a, 1, out); }
int arkmpc_share_sum(arkmpc_ctx* ctx, size_t n, const uint64_t* shares, uint64_t* out_share) { return scalar_reduce(ctx, OP_ADD, n, shares, 2, out_share); }
int arkmpc_scalar_neg(arkmpc_ctx* ctx, size_t n, const uint64_t* a, uint64_t* out) { return scalar_unop(ctx, 0, n, a, out, 32); }
int arkmpc_scalar_from_canonical(arkmpc_ctx* ctx, size_t n, const uint64_t* in, uint64_t* out) { return scalar_unop(ctx, 1, n, in, out, 32); }
int arkmpc_scalar_to_canonical(arkmpc_ctx* ctx, size_t n, const uint64_t* in, uint64_t* out) { return scalar_unop(ctx, 2, n, in, out, 32); }
int arkmpc_scalar_to_bytes_be(arkmpc_ctx* ctx, size_t n, const uint64_t* in, uint8_t* out_bytes) { return scalar_unop(ctx, 3, n, in, out_bytes, 32); }

// ---------------------------------------------------------------------------------------------
// C ABI: ScalarShare vectors
// ---------------------------------------------------------------------------------------------
static int share_binop(arkmpc_ctx* ctx, int op, size_t n, const uint64_t* a, const uint64_t* b, uint64_t* out) {
    ENTER(ctx);
    Stage st(ctx);
    int ia = st.declare_in(a, n * 64), ib = st.declare_in(b, n * 64), io = st.declare_out(out, n * 64);
    st.elementwise(n);
    if (st.commit()) return st.rc;
    size_t lo, cnt;
    while (st.next_chunk(&lo, &cnt)) {
        const dim3 t(TPB);
        const u64 *da = st.in<u64>(ia) + 8 * lo, *db = st.in<u64>(ib) + 8 * lo;
        u64* dout = st.out<u64>(io) + 8 * lo;
        DISPATCH_FIELD(ctx, {
            const dim3 g2(blocks_for(2 * cnt, TPB));              // one thread per field element: 2 per share
            if (op == OP_ADD) hipLaunchKernelGGL((k_scalar_binop<F, OP_ADD>), g2, t, 0, ctx->stream, 2 * cnt, da, db, dout);
            if (op == OP_SUB) hipLaunchKernelGGL((k_scalar_binop<F, OP_SUB>), g2, t, 0, ctx->stream, 2 * cnt, da, db, dout);
        });
    }
    return st.finish();
}
int arkmpc_share_add(arkmpc_ctx* ctx, size_t n, const uint64_t* a, const uint64_t* b, uint64_t* out) { return share_binop(ctx, OP_ADD, n, a, b, out); }
int arkmpc_share_sub(arkmpc_ctx* ctx, size_t n, const uint64_t* a, const uint64_t* b, uint64_t* out) { return share_binop(ctx, OP_SUB, n, a, b, out); }

int arkmpc_share_neg(arkmpc_ctx* ctx, size_t n, const uint64_t* a, uint64_t* out) {
    ENTER(ctx);
    Stage st(ctx);
    int ia = st.declare_in(a, n * 64), io = st.declare_out(out, n * 64);
    st.elementwise(n);
    if (st.commit()) return st.rc;
    size_t lo, cnt;
    while (st.next_chunk(&lo, &cnt))
        DISPATCH_FIELD(ctx, hipLaunchKernelGGL((k_scalar_neg<F>), dim3(blocks_for(2 * cnt, TPB)), dim3(TPB), 0, ctx->stream, 2 * cnt, st.in<u64>(ia) + 8 * lo, st.out<u64>(io) + 8 * lo));
    return st.finish();
}
int arkmpc_share_split(arkmpc_ctx* ctx, size_t n, const uint64_t* aos, uint64_t* out_share_col, uint64_t* out_mac_col) {
    ENTER(ctx);
    Stage st(ctx);
    int ia = st.declare_in(aos, n * 64), is = st.declare_out(out_share_col, n * 32), im = st.declare_out(out_mac_col, n * 32);
    st.elementwise(n);
    if (st.commit()) return st.rc;
    size_t lo, cnt;
    while (st.next_chunk(&lo, &cnt))
        hipLaunchKernelGGL(k_share_split, dim3(blocks_for(cnt, TPB)), dim3(TPB), 0, ctx->stream, cnt, st.in<u64>(ia) + 8 * lo, st.out<u64>(is) + 4 * lo, st.out<u64>(im) + 4 * lo);
    return st.finish();
}
int arkmpc_share_join(arkmpc_ctx* ctx, size_t n, const uint64_t* share_col, const uint64_t* mac_col, uint64_t* out_aos) {
    ENTER(ctx);
    Stage st(ctx);
    int is = st.declare_in(share_col, n * 32), im = st.declare_in(mac_col, n * 32), io = st.declare_out(out_aos, n * 64);
    st.elementwise(n);
    if (st.commit()) return st.rc;
    size_t lo, cnt;
    while (st.next_chunk(&lo, &cnt))
        hipLaunchKernelGGL(k_share_join, dim3(blocks_for(cnt, TPB)), dim3(TPB), 0, ctx->stream, cnt, st.in<u64>(is) + 4 * lo, st.in<u64>(im) + 4 * lo, st.out<u64>(io) + 8 * lo);
    return st.finish();
}
int arkmpc_fill(arkmpc_ctx* ctx, size_t n, size_t words, const uint64_t* record, uint64_t* out) {
    ENTER(ctx);
    if (!record || words == 0 || words > 12 || (words & 1)) return ark_bad(ctx, "fill: record of 2, 4, ..., 12 u64 words");
    Stage st(ctx);
    int io = st.declare_out(out, n * words * 8);
    if (st.commit()) return st.rc;
    FillRecord rec;
    memset(&rec, 0, sizeof(rec));
    memcpy(&rec, record, words * 8);
    const u32 w = (u32)(words / 2);
    if (n) hipLaunchKernelGGL(k_fill_records, dim3(blocks_for(n * w, TPB)), dim3(TPB), 0, ctx->stream, n, w, rec, st.out<uint4>(io));
    return st.finish();
}
int arkmpc_share_extract(arkmpc_ctx* ctx, size_t n, const uint64_t* shares, uint64_t* out) {
    ENTER(ctx);
    Stage st(ctx);
    int ia = st.declare_in(shares, n * 64), io = st.declare_out(out, n * 32);
    st.elementwise(n);
    if (st.commit()) return st.rc;
    size_t lo, cnt;
    while (st.next_chunk(&lo, &cnt))
        DISPATCH_FIELD(ctx, hipLaunchKernelGGL((k_share_extract<F>), dim3(blocks_for(cnt, TPB)), dim3(TPB), 0, ctx->stream, cnt, st.in<u64>(ia) + 8 * lo, st.out<u64>(io) + 4 * lo));
    return st.finish();
}

static int share_addsub_public(arkmpc_ctx* ctx, bool sub, size_t n, int party, const uint64_t key[4], const uint64_t* a,
                               const uint64_t* pub, uint64_t* out) {
    ENTER(ctx);
    if (!party_ok(party)) return ark_bad(ctx, "party_id must be 0 or 1");
    if (!key) return ark_bad(ctx, "null mac_key");
    Stage st(ctx);
    int ia = st.declare_in(a, n * 64), ip = st.declare_in(pub, n * 32), io = st.declare_out(out, n * 64);
    st.elementwise(n);
    if (st.commit()) return st.rc;
    size_t lo, cnt;
    while (st.next_chunk(&lo, &cnt)) {
        Fe k = fe_from_host(key);
        const dim3 t(TPB);
        const u64 *da = st.in<u64>(ia) + 8 * lo, *dp = st.in<u64>(ip) + 4 * lo;
        u64* dout = st.out<u64>(io) + 8 * lo;
        DISPATCH_FIELD(ctx, {
            const dim3 g2(blocks_for(2 * cnt, TPB));
            if (sub) hipLaunchKernelGGL((k_share_addsub_public_flat<F, true>), g2, t, 0, ctx->stream, 2 * cnt, party, k, da, dp, dout);
            else hipLaunchKernelGGL((k_share_addsub_public_flat<F, false>), g2, t, 0, ctx->stream, 2 * cnt, party, k, da, dp, dout);
        });
    }
    return st.finish();
}
int arkmpc_share_add_public(arkmpc_ctx* ctx, size_t n, int party_id, const uint64_t mac_key[4], const uint64_t* a, const uint64_t* pub, uint64_t* out) {
    return share_addsub_public(ctx, false, n, party_id, mac_key, a, pub, out);
}
int arkmpc_share_sub_public(arkmpc_ctx* ctx, size_t n, int party_id, const uint64_t mac_key[4], const uint64_t* a, const uint64_t* pub, uint64_t* out) {
    return share_addsub_public(ctx, true, n, party_id, mac_key, a, pub, out);
}
int arkmpc_share_mul_public(arkmpc_ctx* ctx, size_t n, const uint64_t* a, const uint64_t* pub, uint64_t* out) {
    ENTER(ctx);
    Stage st(ctx);
    int ia = st.declare_in(a, n * 64), ip = st.declare_in(pub, n * 32), io = st.declare_out(out, n * 64);
    st.elementwise(n);
    if (st.commit()) return st.rc;
    size_t lo, cnt;
    while (st.next_chunk(&lo, &cnt))
        DISPATCH_FIELD(ctx, hipLaunchKernelGGL((k_share_mul_public_flat<F>), dim3(blocks_for(2 * cnt, TPB)), dim3(TPB), 0, ctx->stream, 2 * cnt, st.in<u64>(ia) + 8 * lo,
                                               st.in<u64>(ip) + 4 * lo, st.out<u64>(io) + 8 * lo));
    return st.finish();
}

// ---------------------------------------------------------------------------------------------
// C ABI: Beaver multiplication
// ---------------------------------------------------------------------------------------------
// element stride of a column view in u64 units; 0 = broadcast: every element reads the one record at the base pointer (the constant
// batches of a preprocessing source -- `vec![share; n]`, offline_prep.rs:137-158 -- without materialising n copies)
static bool stride_ok(size_t s) { return s == 0 || (s >= 4 && (s % 2) == 0 && s <= 0xffffffffu); }

static int share_public_v(arkmpc_ctx* ctx, int op, size_t n, int party, const uint64_t key[4], const uint64_t* a_share, const uint64_t* a_mac, size_t a_stride,
                          const uint64_t* pub, uint64_t* out_share, uint64_t* out_mac, size_t out_stride) {
    ENTER(ctx);
    if (op != OP_MUL && !party_ok(party)) return ark_bad(ctx, "party_id must be 0 or 1");
    if (op != OP_MUL && !key) return ark_bad(ctx, "null mac_key");
    if (!stride_ok(a_stride) || !stride_ok(out_stride) || !out_stride) return ark_bad(ctx, "bad stride");
    if (ctx->host_buffers) return ark_bad(ctx, "share-view entry points take device pointers only");
    if (n && (!a_share || !a_mac || !pub || !out_share || !out_mac)) return ark_bad(ctx, "null pointer");
    if (((uintptr_t)a_share | (uintptr_t)a_mac | (uintptr_t)pub | (uintptr_t)out_share | (uintptr_t)out_mac) & 15) return ark_bad(ctx, "device pointer not 16-byte aligned");
    if (n) {
        const Fe k = key ? fe_from_host(key) : Fe{};
        dim3 g(blocks_for(n, TPB)), t(TPB);
        const Col as{a_share, (u32)a_stride}, am{a_mac, (u32)a_stride};
        const ColOut os{out_share, (u32)out_stride}, om{out_mac, (u32)out_stride};
        DISPATCH_FIELD(ctx, {
            if (op == OP_ADD) hipLaunchKernelGGL((k_share_public_v<F, OP_ADD>), g, t, 0, ctx->stream, n, party, k, as, am, pub, os, om);
            else if (op == OP_SUB) hipLaunchKernelGGL((k_share_public_v<F, OP_SUB>), g, t, 0, ctx->stream, n, party, k, as, am, pub, os, om);
            else hipLaunchKernelGGL((k_share_public_v<F, OP_MUL>), g, t, 0, ctx->stream, n, party, k, as, am, pub, os, om);
        });
        ARK_HIP(ctx, hipGetLastError());
    }
    return ARKMPC_OK;
}
int arkmpc_share_add_public_v(arkmpc_ctx* ctx, size_t n, int party_id, const uint64_t mac_key[4], const uint64_t* a_share, const uint64_t* a_mac, size_t a_stride,
                              const uint64_t* pub, uint64_t* out_share, uint64_t* out_mac, size_t out_stride) {
    return share_public_v(ctx, OP_ADD, n, party_id, mac_key, a_share, a_mac, a_stride, pub, out_share, out_mac, out_stride);
}
int arkmpc_share_sub_public_v(arkmpc_ctx* ctx, size_t n, int party_id, const uint64_t mac_key[4], const uint64_t* a_share, const uint64_t* a_mac, size_t a_stride,
                              const uint64_t* pub, uint64_t* out_share, uint64_t* out_mac, size_t out_stride) {
    return share_public_v(ctx, OP_SUB, n, party_id, mac_key, a_share, a_mac, a_stride, pub, out_share, out_mac, out_stride);
}
int arkmpc_share_mul_public_v(arkmpc_ctx* ctx, size_t n, const uint64_t* a_share, const uint64_t* a_mac, size_t a_stride, const uint64_t* pub,
                              uint64_t* out_share, uint64_t* out_mac, size_t out_stride) {
    return share_public_v(ctx, OP_MUL, n, 0, nullptr, a_share, a_mac, a_stride, pub, out_share, out_mac, out_stride);
}

int arkmpc_beaver_mask_v(arkmpc_ctx* ctx, size_t n, const uint64_t* x_share, size_t x_stride, const uint64_t* y_share,
                         size_t y_stride, const uint64_t* a_share, size_t a_stride, const uint64_t* b_share, size_t b_stride,
                         uint64_t* out_de) {
    ENTER(ctx);
    if (!stride_ok(x_stride) || !stride_ok(y_stride) || !stride_ok(a_stride) || !stride_ok(b_stride)) return ark_bad(ctx, "bad stride");
    if (ctx->host_buffers) return ark_bad(ctx, "share-view entry points take device pointers only");
    if (n && (!x_share || !y_share || !a_share || !b_share || !out_de)) return ark_bad(ctx, "null pointer");
    if (((uintptr_t)x_share | (uintptr_t)y_share | (uintptr_t)a_share | (uintptr_t)b_share | (uintptr_t)out_de) & 15) return ark_bad(ctx, "device pointer not 16-byte aligned");
    if (n) {
        dim3 g(blocks_for(n, TPB)), t(TPB);
        Col x{x_share, (u32)x_stride}, y{y_share, (u32)y_stride}, a{a_share, (u32)a_stride}, b{b_share, (u32)b_stride};
        (void)g; (void)t;
        DISPATCH_FIELD(ctx, launch_mask<F>(ctx, n, x, y, a, b, out_de));
        hipError_t le = hipGetLastError();
        if (le != hipSuccess) { ark_set_err(ctx, hipGetErrorString(le)); return ARKMPC_ERR_HIP; }
    }
    return ARKMPC_OK;
}

int arkmpc_beaver_mask(arkmpc_ctx* ctx, size_t n, const uint64_t* x, const uint64_t* y, const uint64_t* a, const uint64_t* b, uint64_t* out_de) {
    ENTER(ctx);
    Stage st(ctx);
    int ix = st.declare_in(x, n * 64), iy = st.declare_in(y, n * 64), ia = st.declare_in(a, n * 64), ib = st.declare_in(b, n * 64);
    int io = st.declare_out(out_de, 2 * n * 32, 2);            // two segments of n Scalars: d, then e
    st.elementwise(n);
    if (st.commit()) return st.rc;
    size_t lo, cnt;
    while (st.next_chunk(&lo, &cnt)) {
        Col cx{st.in<u64>(ix) + 8 * lo, 8}, cy{st.in<u64>(iy) + 8 * lo, 8}, ca{st.in<u64>(ia) + 8 * lo, 8}, cb{st.in<u64>(ib) + 8 * lo, 8};
        DISPATCH_FIELD(ctx, launch_mask<F>(ctx, cnt, cx, cy, ca, cb, st.out<u64>(io) + 4 * lo, st.out<u64>(io) + 4 * (n + lo)));
    }
    return st.finish();
}

int arkmpc_beaver_finish(arkmpc_ctx* ctx, size_t n, int party_id, const uint64_t mac_key[4], const uint64_t* d, const uint64_t* e,
                         const uint64_t* a, const uint64_t* b, const uint64_t* c, uint64_t* out) {
    ENTER(ctx);
    if (!party_ok(party_id)) return ark_bad(ctx, "party_id must be 0 or 1");
    if (!mac_key) return ark_bad(ctx, "null mac_key");
    Stage st(ctx);
    int id = st.declare_in(d, n * 32), ie = st.declare_in(e, n * 32);
    int ia = st.declare_in(a, n * 64), ib = st.declare_in(b, n * 64), ic = st.declare_in(c, n * 64), io = st.declare_out(out, n * 64);
    st.elementwise(n);
    if (st.commit()) return st.rc;
    size_t lo, cnt;
    while (st.next_chunk(&lo, &cnt)) {
        Fe k = fe_from_host(mac_key);
        dim3 g(blocks_for(cnt, TPB)), t(TPB);
        const u64 *pa = st.in<u64>(ia) + 8 * lo, *pb = st.in<u64>(ib) + 8 * lo, *pc = st.in<u64>(ic) + 8 * lo;
        u64* po = st.out<u64>(io) + 8 * lo;
        DISPATCH_FIELD(ctx, hipLaunchKernelGGL((k_beaver_finish<F, false>), g, t, 0, ctx->stream, cnt, party_id, k, st.in<u64>(id) + 4 * lo,
                                               st.in<u64>(ie) + 4 * lo, (const u64*)nullptr, (const u64*)nullptr, Col{pa, 8}, Col{pa + 4, 8}, Col{pb, 8}, Col{pb + 4, 8},
                                               Col{pc, 8}, Col{pc + 4, 8}, ColOut{po, 8}, ColOut{po + 4, 8}));
    }
    return st.finish();
}

int arkmpc_beaver_finish_fused(arkmpc_ctx* ctx, size_t n, int party_id, const uint64_t mac_key[4], const uint64_t* my_de,
                               const uint64_t* peer_de, const uint64_t* a, const uint64_t* b, const uint64_t* c, uint64_t* out) {
    ENTER(ctx);
    if (!party_ok(party_id)) return ark_bad(ctx, "party_id must be 0 or 1");
    if (!mac_key) return ark_bad(ctx, "null mac_key");
    Stage st(ctx);
    int i0 = st.declare_in(my_de, 2 * n * 32, 2), i1 = st.declare_in(peer_de, 2 * n * 32, 2);
    int ia = st.declare_in(a, n * 64), ib = st.declare_in(b, n * 64), ic = st.declare_in(c, n * 64), io = st.declare_out(out, n * 64);
    st.elementwise(n);
    if (st.commit()) return st.rc;
    size_t lo, cnt;
    while (st.next_chunk(&lo, &cnt)) {
        Fe k = fe_from_host(mac_key);
        const u64 *pa = st.in<u64>(ia) + 8 * lo, *pb = st.in<u64>(ib) + 8 * lo, *pc = st.in<u64>(ic) + 8 * lo;
        u64* po = st.out<u64>(io) + 8 * lo;
        const u64 *myde = st.in<u64>(i0), *prde = st.in<u64>(i1);
        DISPATCH_FIELD(ctx, launch_finish_fused<F>(ctx, cnt, party_id, k, myde + 4 * lo, prde + 4 * lo, Col{pa, 8}, Col{pa + 4, 8}, Col{pb, 8},
                                                   Col{pb + 4, 8}, Col{pc, 8}, Col{pc + 4, 8}, ColOut{po, 8}, ColOut{po + 4, 8}, myde + 4 * (n + lo), prde + 4 * (n + lo)));
    }
    return st.finish();
}

int arkmpc_beaver_finish_fused_v(arkmpc_ctx* ctx, size_t n, int party_id, const uint64_t mac_key[4], const uint64_t* my_de,
                                 const uint64_t* peer_de, const uint64_t* a_share, const uint64_t* a_mac, size_t a_stride,
                                 const uint64_t* b_share, const uint64_t* b_mac, size_t b_stride, const uint64_t* c_share,
                                 const uint64_t* c_mac, size_t c_stride, uint64_t* out_share, uint64_t* out_mac, size_t out_stride) {
    ENTER(ctx);
    if (!party_ok(party_id)) return ark_bad(ctx, "party_id must be 0 or 1");
    if (!mac_key) return ark_bad(ctx, "null mac_key");
    if (!stride_ok(a_stride) || !stride_ok(b_stride) || !stride_ok(c_stride) || !stride_ok(out_stride) || !out_stride) return ark_bad(ctx, "bad stride");
    if (ctx->host_buffers) return ark_bad(ctx, "share-view entry points take device pointers only");
    if (n && (!my_de || !peer_de || !a_share || !a_mac || !b_share || !b_mac || !c_share || !c_mac || !out_share || !out_mac)) return ark_bad(ctx, "null pointer");
    if (((uintptr_t)my_de | (uintptr_t)peer_de | (uintptr_t)a_share | (uintptr_t)a_mac | (uintptr_t)b_share | (uintptr_t)b_mac |
         (uintptr_t)c_share | (uintptr_t)c_mac | (uintptr_t)out_share | (uintptr_t)out_mac) & 15)
        return ark_bad(ctx, "device pointer not 16-byte aligned");
    if (n) {
        Fe k = fe_from_host(mac_key);
        dim3 g(blocks_for(n, TPB)), t(TPB);
        (void)g; (void)t;
        DISPATCH_FIELD(ctx, launch_finish_fused<F>(ctx, n, party_id, k, my_de, peer_de, Col{a_share, (u32)a_stride}, Col{a_mac, (u32)a_stride},
                                                   Col{b_share, (u32)b_stride}, Col{b_mac, (u32)b_stride}, Col{c_share, (u32)c_stride},
                                                   Col{c_mac, (u32)c_stride}, ColOut{out_share, (u32)out_stride}, ColOut{out_mac, (u32)out_stride}));
        hipError_t le = hipGetLastError();
        if (le != hipSuccess) { ark_set_err(ctx, hipGetErrorString(le)); return ARKMPC_ERR_HIP; }
    }
    return ARKMPC_OK;
}

// Range forms: the gates [lo, lo + n) of a larger batch.  d and e are addressed separately, so a shard can write its slice of the
// FULL d||e buffer (d at +4 lo, e at +4 (N + lo)) -- on its own device or, with peer access, straight into another device's
// buffer -- and read the peer party's slice from wherever it lies.  This is what the multi-device group (arkmpc_group.hip) and the
// one-process-per-GPU sharding launch per device.
int arkmpc_beaver_mask_to(arkmpc_ctx* ctx, size_t n, const uint64_t* x_share, size_t x_stride, const uint64_t* y_share, size_t y_stride,
                          const uint64_t* a_share, size_t a_stride, const uint64_t* b_share, size_t b_stride, uint64_t* out_d, uint64_t* out_e) {
    ENTER(ctx);
    if (!stride_ok(x_stride) || !stride_ok(y_stride) || !stride_ok(a_stride) || !stride_ok(b_stride)) return ark_bad(ctx, "bad stride");
    if (ctx->host_buffers) return ark_bad(ctx, "range entry points take device pointers only");
    if (n && (!x_share || !y_share || !a_share || !b_share || !out_d || !out_e)) return ark_bad(ctx, "null pointer");
    if (((uintptr_t)x_share | (uintptr_t)y_share | (uintptr_t)a_share | (uintptr_t)b_share | (uintptr_t)out_d | (uintptr_t)out_e) & 15)
        return ark_bad(ctx, "device pointer not 16-byte aligned");
    if (n) {
        Col x{x_share, (u32)x_stride}, y{y_share, (u32)y_stride}, a{a_share, (u32)a_stride}, b{b_share, (u32)b_stride};
        DISPATCH_FIELD(ctx, launch_mask<F>(ctx, n, x, y, a, b, out_d, out_e));
        ARK_HIP(ctx, hipGetLastError());
    }
    return ARKMPC_OK;
}
// K1 that writes the payload twice: out_de stays with the party (its own K2+K3 reads it), out_de_msg is the message handed to the link.
// network/mock.rs MOVES a payload to the peer; a device-resident link must not alias a buffer the sender still reads (the two parties'
// streams are not ordered against each other's later reuse), so the alternative is a device-to-device copy after K1 -- one more launch
// in the dependent chain of every round and 128 B per gate of extra traffic.
int arkmpc_beaver_mask_dup(arkmpc_ctx* ctx, size_t n, const uint64_t* x_share, size_t x_stride, const uint64_t* y_share, size_t y_stride,
                           const uint64_t* a_share, size_t a_stride, const uint64_t* b_share, size_t b_stride, uint64_t* out_de, uint64_t* out_de_msg) {
    ENTER(ctx);
    if (!stride_ok(x_stride) || !stride_ok(y_stride) || !stride_ok(a_stride) || !stride_ok(b_stride)) return ark_bad(ctx, "bad stride");
    if (ctx->host_buffers) return ark_bad(ctx, "share-view entry points take device pointers only");
    if (n && (!x_share || !y_share || !a_share || !b_share || !out_de || !out_de_msg)) return ark_bad(ctx, "null pointer");
    if (((uintptr_t)x_share | (uintptr_t)y_share | (uintptr_t)a_share | (uintptr_t)b_share | (uintptr_t)out_de | (uintptr_t)out_de_msg) & 15)
        return ark_bad(ctx, "device pointer not 16-byte aligned");
    if (n) {
        Col x{x_share, (u32)x_stride}, y{y_share, (u32)y_stride}, a{a_share, (u32)a_stride}, b{b_share, (u32)b_stride};
        DISPATCH_FIELD(ctx, launch_mask<F>(ctx, n, x, y, a, b, out_de, nullptr, out_de_msg));
        ARK_HIP(ctx, hipGetLastError());
    }
    return ARKMPC_OK;
}
int arkmpc_beaver_finish_fused_from(arkmpc_ctx* ctx, size_t n, int party_id, const uint64_t mac_key[4], const uint64_t* my_d, const uint64_t* my_e,
                                    const uint64_t* peer_d, const uint64_t* peer_e, const uint64_t* a_share, const uint64_t* a_mac, size_t a_stride,
                                    const uint64_t* b_share, const uint64_t* b_mac, size_t b_stride, const uint64_t* c_share,
                                    const uint64_t* c_mac, size_t c_stride, uint64_t* out_share, uint64_t* out_mac, size_t out_stride) {
    ENTER(ctx);
    if (!party_ok(party_id)) return ark_bad(ctx, "party_id must be 0 or 1");
    if (!mac_key) return ark_bad(ctx, "null mac_key");
    if (!stride_ok(a_stride) || !stride_ok(b_stride) || !stride_ok(c_stride) || !stride_ok(out_stride) || !out_stride) return ark_bad(ctx, "bad stride");
    if (ctx->host_buffers) return ark_bad(ctx, "range entry points take device pointers only");
    if (n && (!my_d || !my_e || !peer_d || !peer_e || !a_share || !a_mac || !b_share || !b_mac || !c_share || !c_mac || !out_share || !out_mac))
        return ark_bad(ctx, "null pointer");
    if (((uintptr_t)my_d | (uintptr_t)my_e | (uintptr_t)peer_d | (uintptr_t)peer_e | (uintptr_t)a_share | (uintptr_t)a_mac | (uintptr_t)b_share |
         (uintptr_t)b_mac | (uintptr_t)c_share | (uintptr_t)c_mac | (uintptr_t)out_share | (uintptr_t)out_mac) & 15)
        return ark_bad(ctx, "device pointer not 16-byte aligned");
    if (n) {
        Fe k = fe_from_host(mac_key);
        DISPATCH_FIELD(ctx, launch_finish_fused<F>(ctx, n, party_id, k, my_d, peer_d, Col{a_share, (u32)a_stride}, Col{a_mac, (u32)a_stride},
                                                   Col{b_share, (u32)b_stride}, Col{b_mac, (u32)b_stride}, Col{c_share, (u32)c_stride},
                                                   Col{c_mac, (u32)c_stride}, ColOut{out_share, (u32)out_stride}, ColOut{out_mac, (u32)out_stride},
                                                   my_e, peer_e));
        ARK_HIP(ctx, hipGetLastError());
    }
    return ARKMPC_OK;
}

// ---------------------------------------------------------------------------------------------
// C ABI: batch open + MAC check
// ---------------------------------------------------------------------------------------------
int arkmpc_mac_check_shares(arkmpc_ctx* ctx, size_t n, const uint64_t mac_key[4], const uint64_t* opened, const uint64_t* shares, uint64_t* out_chk) {
    ENTER(ctx);
    if (!mac_key) return ark_bad(ctx, "null mac_key");
    Stage st(ctx);
    int iv = st.declare_in(opened, n * 32), is = st.declare_in(shares, n * 64), io = st.declare_out(out_chk, n * 32);
    st.elementwise(n);
    if (st.commit()) return st.rc;
    size_t lo, cnt;
    while (st.next_chunk(&lo, &cnt)) {
        Fe k = fe_from_host(mac_key);
        dim3 g(blocks_for(cnt, TPB)), t(TPB);
        DISPATCH_FIELD(ctx, hipLaunchKernelGGL((k_mac_check<F, false>), g, t, 0, ctx->stream, cnt, k, st.in<u64>(iv) + 4 * lo, st.in<u64>(is) + 8 * lo,
                                               (const u64*)nullptr, (u64*)nullptr, st.out<u64>(io) + 4 * lo));
    }
    return st.finish();
}
int arkmpc_open_and_mac_check(arkmpc_ctx* ctx, size_t n, const uint64_t mac_key[4], const uint64_t* shares, const uint64_t* peer,
                              uint64_t* out_opened, uint64_t* out_chk) {
    ENTER(ctx);
    if (!mac_key) return ark_bad(ctx, "null mac_key");
    Stage st(ctx);
    int is = st.declare_in(shares, n * 64), ip = st.declare_in(peer, n * 32);
    int iv = st.declare_out(out_opened, n * 32), io = st.declare_out(out_chk, n * 32);
    st.elementwise(n);
    if (st.commit()) return st.rc;
    size_t lo, cnt;
    while (st.next_chunk(&lo, &cnt)) {
        Fe k = fe_from_host(mac_key);
        dim3 g(blocks_for(cnt, TPB)), t(TPB);
        DISPATCH_FIELD(ctx, hipLaunchKernelGGL((k_mac_check<F, true>), g, t, 0, ctx->stream, cnt, k, (const u64*)nullptr, st.in<u64>(is) + 8 * lo,
                                               st.in<u64>(ip) + 4 * lo, st.out<u64>(iv) + 4 * lo, st.out<u64>(io) + 4 * lo));
    }
    return st.finish();
}
int arkmpc_open_and_mac_check_v(arkmpc_ctx* ctx, size_t n, const uint64_t mac_key[4], const uint64_t* share_col, const uint64_t* mac_col, size_t stride,
                                const uint64_t* peer, uint64_t* out_opened, uint64_t* out_chk) {
    ENTER(ctx);
    if (!mac_key) return ark_bad(ctx, "null mac_key");
    if (!stride_ok(stride)) return ark_bad(ctx, "bad stride");
    if (ctx->host_buffers) return ark_bad(ctx, "share-view entry points take device pointers only");
    if (n && (!share_col || !mac_col || !peer || !out_opened || !out_chk)) return ark_bad(ctx, "null pointer");
    if (((uintptr_t)share_col | (uintptr_t)mac_col | (uintptr_t)peer | (uintptr_t)out_opened | (uintptr_t)out_chk) & 15) return ark_bad(ctx, "device pointer not 16-byte aligned");
    if (n) {
        const Fe k = fe_from_host(mac_key);
        dim3 g(blocks_for(n, TPB)), t(TPB);
        DISPATCH_FIELD(ctx, launch_k(ctx, k_mac_check_v<F>, g, t, n, k, Col{share_col, (u32)stride}, Col{mac_col, (u32)stride}, peer, out_opened, out_chk));
        ARK_HIP(ctx, hipGetLastError());
    }
    return ARKMPC_OK;
}
// enqueue K5 on the context's stream; failures accumulate in the sticky verify flag
static int mac_verify_enqueue(arkmpc_ctx* ctx, size_t n, const uint64_t* mine, const uint64_t* peer, bool keep_staged) {
    Stage st(ctx);
    int im = st.declare_in(mine, n * 32), ip = st.declare_in(peer, n * 32);
    if (st.commit()) return st.rc;
    if (n) {
        dim3 g(blocks_for(n, TPB)), t(TPB);
        DISPATCH_FIELD(ctx, launch_k(ctx, k_mac_verify<F>, g, t, n, st.in<u64>(im), st.in<u64>(ip), ctx->d_vflag, ctx->d_vgate));
    }
    ARK_HIP(ctx, hipGetLastError());
    // host-buffer mode stages through the arena, which the next staged call reuses: the kernel must have consumed it first
    if (ctx->host_buffers && !keep_staged) ARK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ARKMPC_OK;
}
// after the stream has drained: read and clear the sticky flag; a failure also re-opens the device-side gate
static int mac_verify_collect(arkmpc_ctx* ctx, int* out_ok) {
    const int failed = __atomic_load_n(ctx->h_vflag, __ATOMIC_ACQUIRE);
    *out_ok = failed == 0 ? 1 : 0;
    if (failed) {
        // re-open the device-side gate FIRST: were the host flag cleared and the memset then to fail, the gate would stay shut with the
        // flag at 0 and every later failed verification on this context would go unreported (fail-open).  On error the flag stays set.
        ARK_HIP(ctx, hipMemsetAsync(ctx->d_vgate, 0, sizeof(int), ctx->stream));
        ARK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        __atomic_store_n(ctx->h_vflag, 0, __ATOMIC_RELEASE);
    }
    return ARKMPC_OK;
}
int arkmpc_mac_verify_async(arkmpc_ctx* ctx, size_t n, const uint64_t* mine, const uint64_t* peer) {
    ENTER(ctx);
    return mac_verify_enqueue(ctx, n, mine, peer, false);
}
int arkmpc_mac_verify_result(arkmpc_ctx* ctx, int* out_ok) {
    ENTER(ctx);
    if (!out_ok) return ark_bad(ctx, "null out_ok");
    ARK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return mac_verify_collect(ctx, out_ok);
}
int arkmpc_mac_verify(arkmpc_ctx* ctx, size_t n, const uint64_t* mine, const uint64_t* peer, int* out_ok) {
    ENTER(ctx);
    if (!out_ok) return ark_bad(ctx, "null out_ok");
    int rc = mac_verify_enqueue(ctx, n, mine, peer, true);
    if (rc) return rc;
    ARK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return mac_verify_collect(ctx, out_ok);                                        // includes failures of earlier _async calls not yet collected
}

// H1: commitment.rs:63-89 / :30-43.  K6 on the GPU in chunks; D2H into pinned double buffers; the
// sponge absorbs chunk c on the host while chunk c+1 is converted and copied.
int arkmpc_commit_sha3(arkmpc_ctx* ctx, size_t n, const uint64_t* values, const uint64_t blinder[4], uint64_t out_commitment[4]) {
    ENTER(ctx);
    if (!blinder || !out_commitment) return ark_bad(ctx, "null blinder/out");
    if (n && !values) return ark_bad(ctx, "null values");
    const size_t CHUNK = (size_t)1 << 18;  // elements per chunk (8 MiB of bytes)
    if (!ctx->h_pin[0]) {
        for (int i = 0; i < 2; ++i) {
            ARK_HIP(ctx, hipHostMalloc((void**)&ctx->h_pin[i], CHUNK * 32));
            ARK_HIP(ctx, hipEventCreateWithFlags(&ctx->ev[i], hipEventDisableTiming));
        }
        ctx->h_pin_cap = CHUNK * 32;
    }
    // device staging: [2 x CHUNK*32 bytes out] (+ [2 x CHUNK*32 in] in host-buffer mode)
    const size_t need = 2 * CHUNK * 32 * (ctx->host_buffers ? 2 : 1);
    int rc = arena_reserve(ctx, need);
    if (rc) return rc;
    if (!ctx->host_buffers && ((uintptr_t)values & 15)) return ark_bad(ctx, "device pointer not 16-byte aligned");
    unsigned char* d_out[2] = {(unsigned char*)ctx->arena, (unsigned char*)ctx->arena + CHUNK * 32};
    u64* d_in[2] = {(u64*)(ctx->arena + 2 * CHUNK * 32), (u64*)(ctx->arena + 3 * CHUNK * 32)};
    Sha3State sh;
    sha3_256_init(&sh);
    const size_t nchunks = (n + CHUNK - 1) / CHUNK;
    auto issue = [&](size_t c) -> int {
        const size_t off = c * CHUNK, cnt = (n - off < CHUNK) ? (n - off) : CHUNK;
        const int s = (int)(c & 1);
        const u64* src = values + 4 * off;
        if (ctx->host_buffers) {
            ARK_HIP(ctx, hipMemcpyAsync(d_in[s], src, cnt * 32, hipMemcpyHostToDevice, ctx->stream));
            src = d_in[s];
        }
        dim3 g(blocks_for(cnt, TPB)), t(TPB);
        DISPATCH_FIELD(ctx, hipLaunchKernelGGL((k_to_bytes_be<F>), g, t, 0, ctx->stream, cnt, src, d_out[s]));
        ARK_HIP(ctx, hipGetLastError());
        ARK_HIP(ctx, hipMemcpyAsync(ctx->h_pin[s], d_out[s], cnt * 32, hipMemcpyDeviceToHost, ctx->stream));
        ARK_HIP(ctx, hipEventRecord(ctx->ev[s], ctx->stream));
        return ARKMPC_OK;
    };
    if (nchunks) { rc = issue(0); if (rc) return rc; }
    for (size_t c = 0; c < nchunks; ++c) {
        const int s = (int)(c & 1);
        ARK_HIP(ctx, hipEventSynchronize(ctx->ev[s]));
        // slot s^1 was fully absorbed in the previous iteration, so it is free for chunk c+1
        if (c + 1 < nchunks) { rc = issue(c + 1); if (rc) return rc; }
        const size_t off = c * CHUNK, cnt = (n - off < CHUNK) ? (n - off) : CHUNK;
        sha3_256_update(&sh, ctx->h_pin[s], cnt * 32);
    }
    unsigned char be[32], dig[32];
    host_to_bytes_be(ctx->field_id, blinder, be);
    sha3_256_update(&sh, be, 32);
    sha3_256_final(&sh, dig);
    host_from_be_bytes_mod_order(ctx->field_id, dig, out_commitment);
    return ARKMPC_OK;
}

}  // extern "C"
#include "arkmpc_stream.inc"
extern "C" {

int arkmpc_sha3_256(const uint8_t* msg, size_t len, uint8_t out32[32]) {
    if ((!msg && len) || !out32) return ARKMPC_ERR_BAD_ARG;
    Sha3State sh;
    sha3_256_init(&sh);
    sha3_256_update(&sh, msg, len);
    sha3_256_final(&sh, out32);
    return ARKMPC_OK;
}

}  // extern "C"
