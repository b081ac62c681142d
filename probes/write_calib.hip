// WRITE_SIZE calibration (not product code): pure store kernels with a KNOWN byte count in the store patterns of the path, so that
// rocprofv3's WRITE_SIZE (uncalibrated on gfx950 per MI355X_MICROARCH.md) can be read against them:
//   pair      one thread = one 32-byte element as two 16-byte stores (lanes 32 B apart): the per-column pattern of K1 / K2+K3 / K4
//   k3        the exact store pattern of k_beaver_finish_asm in the split layout: share column + MAC column, 4 x 16 B per thread
//   line      one thread = 16 bytes, lanes contiguous (a wave instruction writes 16 whole 64-byte lines)
//   quad      the 32-byte elements of `pair` re-dealt inside lane quads so that every store INSTRUCTION writes whole 64-byte lines
// each with plain and non-temporal stores.  Every kernel writes exactly n_elems * 32 bytes (k3: * 64).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 probes/write_calib.hip -o probes/write_calib
// Run:   probes/write_calib [log2_elems=21] [reps=20]      (prints one JSON line per kernel: bytes per launch, us per launch)
//        rocprofv3 --pmc WRITE_SIZE --kernel-trace -f csv -d <dir> -- probes/write_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
typedef unsigned int u32;
typedef u32 v4u __attribute__((ext_vector_type(4)));

template <int NT> __device__ inline void st(v4u* p, v4u v) {
    if (NT) __builtin_nontemporal_store(v, p); else *p = v;
}

template <int NT> __global__ void __launch_bounds__(256) k_pair(size_t n, v4u* out, u32 seed) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const v4u a = {seed + (u32)i, seed, 1u, 2u}, b = {seed ^ (u32)i, 3u, 4u, seed};
    st<NT>(out + 2 * i, a);
    st<NT>(out + 2 * i + 1, b);
}
template <int NT> __global__ void __launch_bounds__(256) k_k3(size_t n, v4u* out_s, v4u* out_m, u32 seed) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const v4u a = {seed + (u32)i, seed, 1u, 2u}, b = {seed ^ (u32)i, 3u, 4u, seed};
    st<NT>(out_s + 2 * i, a);
    st<NT>(out_s + 2 * i + 1, b);
    st<NT>(out_m + 2 * i, b);
    st<NT>(out_m + 2 * i + 1, a);
}
template <int NT> __global__ void __launch_bounds__(256) k_line(size_t n16, v4u* out, u32 seed) {     // n16 = number of 16-byte vectors
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n16) return;
    const v4u a = {seed + (u32)i, seed, 1u, 2u};
    st<NT>(out + i, a);
}
// lanes 4q..4q+3 own elements 4q..4q+3 (32 B each = lines 2q and 2q+1 hold two elements each).  Instruction 1 writes line 2q... of every
// quad: lane 4q+k stores chunk k of the 64-byte line made of elements 4q and 4q+1; instruction 2 the line of elements 4q+2, 4q+3.
template <int NT> __global__ void __launch_bounds__(256) k_quad(size_t n, v4u* out, u32 seed) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;                                                  // n is a multiple of 256 here
    const v4u lo = {seed + (u32)i, seed, 1u, 2u}, hi = {seed ^ (u32)i, 3u, 4u, seed};
    const u32 k = threadIdx.x & 3;
    // chunk k of line A (elements 4q, 4q+1): element (k >> 1) of the quad, half (k & 1); line B: element 2 + (k >> 1)
    v4u va, vb;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const u32 l = lo[c], h = hi[c];
        const u32 srcA = (threadIdx.x & ~3u) + (k >> 1), srcB = srcA + 2;
        const u32 la = __shfl(l, srcA), ha = __shfl(h, srcA), lb = __shfl(l, srcB), hb = __shfl(h, srcB);
        va[c] = (k & 1) ? ha : la;
        vb[c] = (k & 1) ? hb : lb;
    }
    const size_t q = i >> 2;
    st<NT>(out + 8 * q + k, va);
    st<NT>(out + 8 * q + 4 + k, vb);
}

int main(int argc, char** argv) {
    const int lg = argc > 1 ? atoi(argv[1]) : 21;
    const int reps = argc > 2 ? atoi(argv[2]) : 20;
    const size_t n = (size_t)1 << lg;
    v4u *a = nullptr, *b = nullptr;
    CK(hipMalloc((void**)&a, n * 32 + 64));
    CK(hipMalloc((void**)&b, n * 32 + 64));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const dim3 g((unsigned)(n / 256)), g16((unsigned)(2 * n / 256)), t(256);
    struct Case { const char* name; int id; size_t bytes; };
    const Case cases[] = {{"pair", 0, n * 32}, {"pair_nt", 1, n * 32}, {"k3", 2, n * 64}, {"k3_nt", 3, n * 64}, {"line", 4, n * 32}, {"line_nt", 5, n * 32},
                          {"quad", 6, n * 32}, {"quad_nt", 7, n * 32}};
    for (const Case& c : cases) {
        float best = 1e30f, sum = 0;
        for (int r = 0; r < reps + 2; ++r) {
            CK(hipEventRecord(e0, 0));
            switch (c.id) {
                case 0: hipLaunchKernelGGL(k_pair<0>, g, t, 0, 0, n, a, (u32)r); break;
                case 1: hipLaunchKernelGGL(k_pair<1>, g, t, 0, 0, n, a, (u32)r); break;
                case 2: hipLaunchKernelGGL(k_k3<0>, g, t, 0, 0, n, a, b, (u32)r); break;
                case 3: hipLaunchKernelGGL(k_k3<1>, g, t, 0, 0, n, a, b, (u32)r); break;
                case 4: hipLaunchKernelGGL(k_line<0>, g16, t, 0, 0, 2 * n, a, (u32)r); break;
                case 5: hipLaunchKernelGGL(k_line<1>, g16, t, 0, 0, 2 * n, a, (u32)r); break;
                case 6: hipLaunchKernelGGL(k_quad<0>, g, t, 0, 0, n, a, (u32)r); break;
                case 7: hipLaunchKernelGGL(k_quad<1>, g, t, 0, 0, n, a, (u32)r); break;
            }
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (r >= 2) { sum += ms; if (ms < best) best = ms; }
        }
        printf("{\"kernel\": \"%s\", \"bytes_per_launch\": %zu, \"avg_us\": %.2f, \"best_us\": %.2f, \"GBps_avg\": %.1f}\n", c.name, c.bytes, sum / reps * 1e3,
               best * 1e3, c.bytes / (sum / reps * 1e-3) / 1e9);
    }
    CK(hipFree(a)); CK(hipFree(b));
    return 0;
}
