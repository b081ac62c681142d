// Probe for the design choice "one thread per element" vs the limb-sliced form BASELINE.json's north_star sketches
// (SoA limbs, one limb per lane, carries moved between lanes with wavefront shuffles): the same batch of 256-bit
// Montgomery multiplications over BN254 Fr computed both ways, results compared, both timed.
//   thread-per-element : fp.hpp's fe_mul (8 x u32 limbs in 8 VGPRs of one lane, carries in VCC)
//   limb-sliced        : 8 adjacent lanes own one element (lane l holds limb l); CIOS rows with the multiplier limb
//                        broadcast by ds_bpermute / DPP shuffles, per-lane 64-bit lazy accumulators, the limb shift and
//                        the carries moved one lane up with shuffles, final carry / borrow resolution across the 8 lanes.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I ark-mpc_amd/csrc probes/limb_sliced.hip -o ark-mpc_amd/lib/limb_sliced
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include "fp.hpp"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
constexpr int F = 0;   // BN254 Fr

__global__ void __launch_bounds__(256) k_thread_per_element(size_t n, const u64* a, const u64* b, u64* out, int reps) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    Fe x = fe_load(a + 4 * i), y = fe_load(b + 4 * i);
    for (int r = 0; r < reps; ++r) x = fe_mul<F>(x, y);
    fe_store(out + 4 * i, x);
}

// one Montgomery multiplication, 8 lanes per element; returns this lane's limb of the canonical product
__device__ __forceinline__ u32 mont_mul_sliced(u32 a, u32 b, u32 p, u32 inv, int lane8, int base) {
    u64 t = 0;                                           // lazy accumulator: limb value + pending carry
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const u32 ai = __shfl(a, base + i);              // broadcast limb i of a to the group
        u64 acc = t + (u64)ai * b;
        u32 lo = (u32)acc;
        const u32 hi = (u32)(acc >> 32);
        const u32 m = __shfl(lo, base) * inv;            // lane 0's low word decides the reduction multiple
        acc = (u64)lo + (u64)m * p;
        lo = (u32)acc;
        const u32 hi2 = (u32)(acc >> 32);
        u32 up = __shfl_down(lo, 1);                     // limb shift: lane l takes the low word of lane l + 1
        if (lane8 == 7) up = 0;
        t = (u64)up + hi + hi2;
    }
    // carry resolution: every lane passes its pending carry up until none is left (ripples only through 0xffffffff limbs)
    u32 v = (u32)t, c = (u32)(t >> 32), top = 0;
    for (int k = 0; k < 8; ++k) {
        u32 cin = __shfl_up(c, 1);
        if (lane8 == 0) cin = 0;
        if (lane8 == 7) top += c;                        // carry out of the top limb (value in [0, 2p) needs at most 1 bit)
        const u64 s = (u64)v + cin;
        v = (u32)s; c = (u32)(s >> 32);
        if (!__any(c != 0)) break;
    }
    // conditional subtraction of p: borrow chain across the lanes
    u32 d = v, bor = 0;
    {
        // lane-local difference and borrow-generate / propagate flags, then a ripple over 8 lanes
        const u64 s0 = (u64)v - p;
        d = (u32)s0;
        u32 g = (u32)(s0 >> 63);                          // generates a borrow
        u32 pr = (d == 0);                                // would propagate an incoming borrow
        u32 bin = 0;
        for (int k = 0; k < 8; ++k) {
            u32 from_below = __shfl_up(g | (pr & bin), 1);
            if (lane8 == 0) from_below = 0;
            const bool changed = from_below != bin;
            bin = from_below;
            if (k && !__any(changed)) break;
        }
        d -= bin;
        bor = g | (pr & bin);
    }
    const u32 top_all = __shfl(top, base + 7), bor_top = __shfl(bor, base + 7);
    const bool take = top_all || !bor_top;               // t >= p
    return take ? d : v;
}

__global__ void __launch_bounds__(256) k_limb_sliced(size_t n, const u32* a, const u32* b, u32* out, int reps) {
    // element e of the batch = 8 consecutive u32 (AoS limbs): lane l of a group loads limb l -- a fully coalesced access
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t e = t >> 3;
    const int lane8 = threadIdx.x & 7, base = (threadIdx.x & 63) & ~7;
    const bool live = e < n;
    u32 x = live ? a[t] : 0, y = live ? b[t] : 0;
    const u32 p = FieldParams<F>::P(lane8);
    for (int r = 0; r < reps; ++r) x = mont_mul_sliced(x, y, p, FieldParams<F>::INV32, lane8, base);
    if (live) out[t] = x;
}

int main(int argc, char** argv) {
    const size_t n = (size_t)1 << (argc > 1 ? atoi(argv[1]) : 22);
    const int reps = argc > 2 ? atoi(argv[2]) : 16;
    std::vector<u64> ha(4 * n), hb(4 * n);
    u64 s = 0x9E3779B97F4A7C15ull;
    auto nxt = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    for (size_t i = 0; i < n; ++i) for (int k = 0; k < 4; ++k) { ha[4 * i + k] = nxt(); hb[4 * i + k] = nxt(); if (k == 3) { ha[4 * i + 3] >>= 4; hb[4 * i + 3] >>= 4; } }
    u64 *a, *b, *o1, *o2;
    CK(hipMalloc(&a, n * 32)); CK(hipMalloc(&b, n * 32)); CK(hipMalloc(&o1, n * 32)); CK(hipMalloc(&o2, n * 32));
    CK(hipMemcpy(a, ha.data(), n * 32, hipMemcpyHostToDevice)); CK(hipMemcpy(b, hb.data(), n * 32, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms1 = 0, ms2 = 0;
    for (int it = 0; it < 3; ++it) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_thread_per_element, dim3((n + 255) / 256), dim3(256), 0, 0, n, a, b, o1, reps);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms1, e0, e1));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_limb_sliced, dim3((8 * n + 255) / 256), dim3(256), 0, 0, n, (const u32*)a, (const u32*)b, (u32*)o2, reps);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms2, e0, e1));
    }
    std::vector<u64> r1(4 * n), r2(4 * n);
    CK(hipMemcpy(r1.data(), o1, n * 32, hipMemcpyDeviceToHost)); CK(hipMemcpy(r2.data(), o2, n * 32, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (size_t i = 0; i < 4 * n; ++i) bad += r1[i] != r2[i];
    printf("{\"n\": %zu, \"reps\": %d, \"mismatching_words\": %zu, \"thread_per_element_ms\": %.4f, \"limb_sliced_ms\": %.4f, "
           "\"thread_per_element_Gmul_per_s\": %.2f, \"limb_sliced_Gmul_per_s\": %.2f}\n",
           n, reps, bad, ms1, ms2, n * (double)reps / ms1 * 1e-6, n * (double)reps / ms2 * 1e-6);
    return bad ? 1 : 0;
}
