// Microbenchmarks that drive the kernel design (not product code):
//  (1) VALU issue rates of the integer-multiply forms a 256-bit Montgomery mul can be built from
//  (2) HBM streaming rate of three ways to read 64-byte AoS records (arkworks ScalarShare layout)
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 probes/ubench.hip -o probes/ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

typedef unsigned long long u64;
typedef unsigned int u32;

// ---------------------------------------------------------------- instruction rates
template <int OP>
__global__ void __launch_bounds__(256) rate_kernel(u32* out, u32 seed, int iters) {
    u32 a0 = seed + threadIdx.x, a1 = a0 * 3 + 1, a2 = a0 * 5 + 7, a3 = a0 * 7 + 11;
    u32 b0 = seed ^ 0x9e3779b9u, b1 = b0 + 17, b2 = b0 + 31, b3 = b0 + 57;
    u64 c0 = a0, c1 = a1, c2 = a2, c3 = a3, c4 = a0 + 1, c5 = a1 + 1, c6 = a2 + 1, c7 = a3 + 1;
    double d0 = a0, d1 = a1, d2 = a2, d3 = a3, d4 = 1.5, d5 = 2.5, d6 = 3.5, d7 = 4.5;
    const double dm = 1.0000001, da = 0.5;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if constexpr (OP == 0) {  // v_mad_u64_u32, 8 independent chains
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(c0) : "v"(a0), "v"(b0) : "vcc");
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(c1) : "v"(a1), "v"(b1) : "vcc");
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(c2) : "v"(a2), "v"(b2) : "vcc");
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(c3) : "v"(a3), "v"(b3) : "vcc");
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(c4) : "v"(a0), "v"(b1) : "vcc");
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(c5) : "v"(a1), "v"(b2) : "vcc");
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(c6) : "v"(a2), "v"(b3) : "vcc");
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(c7) : "v"(a3), "v"(b0) : "vcc");
            } else if constexpr (OP == 1) {  // v_mul_lo_u32
                u32 *p0 = (u32*)&c0, *p1 = (u32*)&c1, *p2 = (u32*)&c2, *p3 = (u32*)&c3;
                asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(p0[0]) : "v"(b0));
                asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(p0[1]) : "v"(b1));
                asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(p1[0]) : "v"(b2));
                asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(p1[1]) : "v"(b3));
                asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(p2[0]) : "v"(b0));
                asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(p2[1]) : "v"(b1));
                asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(p3[0]) : "v"(b2));
                asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(p3[1]) : "v"(b3));
            } else if constexpr (OP == 2) {  // v_mul_hi_u32
                u32 *p0 = (u32*)&c0, *p1 = (u32*)&c1, *p2 = (u32*)&c2, *p3 = (u32*)&c3;
                asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(p0[0]) : "v"(b0));
                asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(p0[1]) : "v"(b1));
                asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(p1[0]) : "v"(b2));
                asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(p1[1]) : "v"(b3));
                asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(p2[0]) : "v"(b0));
                asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(p2[1]) : "v"(b1));
                asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(p3[0]) : "v"(b2));
                asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(p3[1]) : "v"(b3));
            } else if constexpr (OP == 3) {  // v_add_co_u32 + v_addc_co_u32 pairs (64-bit add), 4 chains
                u32 *p0 = (u32*)&c0, *p1 = (u32*)&c1, *p2 = (u32*)&c2, *p3 = (u32*)&c3;
                asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, %1, %3, vcc" : "+v"(p0[0]), "+v"(p0[1]) : "v"(b0), "v"(b1) : "vcc");
                asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, %1, %3, vcc" : "+v"(p1[0]), "+v"(p1[1]) : "v"(b2), "v"(b3) : "vcc");
                asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, %1, %3, vcc" : "+v"(p2[0]), "+v"(p2[1]) : "v"(b0), "v"(b1) : "vcc");
                asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, %1, %3, vcc" : "+v"(p3[0]), "+v"(p3[1]) : "v"(b2), "v"(b3) : "vcc");
            } else if constexpr (OP == 4) {  // v_fma_f64, 8 chains
                asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d0) : "v"(dm), "v"(da));
                asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d1) : "v"(dm), "v"(da));
                asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d2) : "v"(dm), "v"(da));
                asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d3) : "v"(dm), "v"(da));
                asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d4) : "v"(dm), "v"(da));
                asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d5) : "v"(dm), "v"(da));
                asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d6) : "v"(dm), "v"(da));
                asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d7) : "v"(dm), "v"(da));
            } else if constexpr (OP == 5) {  // v_mad_u32_u24 (full-rate 24-bit mul-add)
                u32 *p0 = (u32*)&c0, *p1 = (u32*)&c1, *p2 = (u32*)&c2, *p3 = (u32*)&c3;
                asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(p0[0]) : "v"(b0), "v"(b1));
                asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(p0[1]) : "v"(b1), "v"(b2));
                asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(p1[0]) : "v"(b2), "v"(b3));
                asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(p1[1]) : "v"(b3), "v"(b0));
                asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(p2[0]) : "v"(b0), "v"(b1));
                asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(p2[1]) : "v"(b1), "v"(b2));
                asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(p3[0]) : "v"(b2), "v"(b3));
                asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(p3[1]) : "v"(b3), "v"(b0));
            } else if constexpr (OP == 6) {  // v_add_u32 plain, 8 chains (full-rate reference)
                u32 *p0 = (u32*)&c0, *p1 = (u32*)&c1, *p2 = (u32*)&c2, *p3 = (u32*)&c3;
                asm volatile("v_add_u32 %0, %0, %1" : "+v"(p0[0]) : "v"(b0));
                asm volatile("v_add_u32 %0, %0, %1" : "+v"(p0[1]) : "v"(b1));
                asm volatile("v_add_u32 %0, %0, %1" : "+v"(p1[0]) : "v"(b2));
                asm volatile("v_add_u32 %0, %0, %1" : "+v"(p1[1]) : "v"(b3));
                asm volatile("v_add_u32 %0, %0, %1" : "+v"(p2[0]) : "v"(b0));
                asm volatile("v_add_u32 %0, %0, %1" : "+v"(p2[1]) : "v"(b1));
                asm volatile("v_add_u32 %0, %0, %1" : "+v"(p3[0]) : "v"(b2));
                asm volatile("v_add_u32 %0, %0, %1" : "+v"(p3[1]) : "v"(b3));
            } else if constexpr (OP == 7) {  // v_mul_hi_u32_u24 + v_mul_u32_u24 pair
                u32 *p0 = (u32*)&c0, *p1 = (u32*)&c1, *p2 = (u32*)&c2, *p3 = (u32*)&c3;
                asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(p0[0]) : "v"(b0));
                asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(p0[1]) : "v"(b1));
                asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(p1[0]) : "v"(b2));
                asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(p1[1]) : "v"(b3));
                asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(p2[0]) : "v"(b0));
                asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(p2[1]) : "v"(b1));
                asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(p3[0]) : "v"(b2));
                asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(p3[1]) : "v"(b3));
            }
        }
    }
    u64 s = c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7;
    double ds = d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7;
    out[blockIdx.x * blockDim.x + threadIdx.x] = (u32)s ^ (u32)(s >> 32) ^ (u32)ds;
}

template <int OP>
int run_rate(const char* name, int ops_per_unroll, u32* dout) {
    const int blocks = 256 * 8, threads = 256, iters = 2000;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    rate_kernel<OP><<<blocks, threads>>>(dout, 1, 10);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    rate_kernel<OP><<<blocks, threads>>>(dout, 1, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double ops = (double)blocks * threads * iters * 16.0 * ops_per_unroll;
    printf("RATE %-28s %8.3f ms  %8.2f Gop/s(lane)  => %.2f cyc/wave-instr/SIMD @2.4GHz\n", name, ms, ops / ms * 1e-6,
           (256.0 * 4 * 64 * 2.4e9) / (ops / (ms * 1e-3)));
    return 0;
}

// ---------------------------------------------------------------- record-load strategies
// Work: out[i] = xor-fold of a 64-byte record i from each of NARR arrays (forces all bytes to be read).
__device__ __forceinline__ uint4 x4(uint4 a, uint4 b) { return make_uint4(a.x ^ b.x, a.y ^ b.y, a.z ^ b.z, a.w ^ b.w); }

// (A) naive: thread i reads its own 64-B record with 4 x dwordx4 (stride 64 B across lanes)
template <int NARR>
__global__ void __launch_bounds__(256) load_strided(const uint4* const* arrs, uint4* out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        uint4 acc = make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int a = 0; a < NARR; ++a) {
            const uint4* p = arrs[a] + i * 4;
            acc = x4(acc, x4(x4(p[0], p[1]), x4(p[2], p[3])));
        }
        out[i] = acc;
    }
}

// (B) per-wave LDS transpose: coalesced dwordx4 loads of 64 records (4 KiB), record stride 80 B in LDS
template <int NARR>
__global__ void __launch_bounds__(256) load_lds(const uint4* const* arrs, uint4* out, size_t n) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[4 * 64 * 80];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned char* my = smem + wave * 64 * 80;
    size_t nwaves = (size_t)gridDim.x * 4;
    for (size_t w = (size_t)blockIdx.x * 4 + wave; w * 64 < n; w += nwaves) {
        size_t base = w * 64;
        uint4 acc = make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int a = 0; a < NARR; ++a) {
            const uint4* p = arrs[a] + base * 4;
            uint4 c0 = p[lane], c1 = p[64 + lane], c2 = p[128 + lane], c3 = p[192 + lane];
            const int wo = (lane >> 2) * 80 + (lane & 3) * 16;
            *(uint4*)(my + wo) = c0;
            *(uint4*)(my + wo + 16 * 80) = c1;
            *(uint4*)(my + wo + 32 * 80) = c2;
            *(uint4*)(my + wo + 48 * 80) = c3;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const uint4* r = (const uint4*)(my + lane * 80);
            uint4 r0 = r[0], r1 = r[1], r2 = r[2], r3 = r[3];
            acc = x4(acc, x4(x4(r0, r1), x4(r2, r3)));
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        out[base + lane] = acc;
    }
}

// (C) planar: 4 planes of 16 B per record -> fully coalesced, no LDS
template <int NARR>
__global__ void __launch_bounds__(256) load_planar(const uint4* const* arrs, uint4* out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        uint4 acc = make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int a = 0; a < NARR; ++a) {
            const uint4* p = arrs[a];
            acc = x4(acc, x4(x4(p[i], p[n + i]), x4(p[2 * n + i], p[3 * n + i])));
        }
        out[i] = acc;
    }
}

// (D) plain float4 copy for calibration of achievable HBM rate
__global__ void __launch_bounds__(256) copy16(const uint4* __restrict__ in, uint4* __restrict__ out, size_t n16) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n16; i += stride) out[i] = in[i];
}

int main() {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("device %s  CUs %d  clock %d kHz  memclock %d kHz  L2 %d\n", prop.name, prop.multiProcessorCount, prop.clockRate, prop.memoryClockRate, prop.l2CacheSize);
    u32* dout; CK(hipMalloc(&dout, 256 * 8 * 256 * 4));
    run_rate<6>("v_add_u32", 8, dout);
    run_rate<0>("v_mad_u64_u32", 8, dout);
    run_rate<1>("v_mul_lo_u32", 8, dout);
    run_rate<2>("v_mul_hi_u32", 8, dout);
    run_rate<3>("v_add_co+v_addc (2 instr)", 8, dout);
    run_rate<4>("v_fma_f64", 8, dout);
    run_rate<5>("v_mad_u32_u24", 8, dout);
    run_rate<7>("v_mul_hi_u32_u24/mul_u32_u24", 8, dout);

    // loads: 5 arrays of 2^22 records (256 MiB each) -> beyond the 256 MiB infinity cache
    const size_t n = (size_t)1 << 22; const int NARR = 5;
    std::vector<uint4*> h(NARR);
    for (int a = 0; a < NARR; ++a) { CK(hipMalloc(&h[a], n * 64)); CK(hipMemset(h[a], 0x5a + a, n * 64)); }
    const uint4** darrs; CK(hipMalloc(&darrs, NARR * sizeof(void*)));
    CK(hipMemcpy(darrs, h.data(), NARR * sizeof(void*), hipMemcpyHostToDevice));
    uint4* out; CK(hipMalloc(&out, n * 16));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double bytes = (double)n * (64.0 * NARR + 16.0);
    for (int grid : {2048, 4096, 16384}) {
        for (int variant = 0; variant < 3; ++variant) {
            float best = 1e9;
            for (int rep = 0; rep < 5; ++rep) {
                CK(hipEventRecord(e0));
                if (variant == 0) load_strided<NARR><<<grid, 256>>>(darrs, out, n);
                if (variant == 1) load_lds<NARR><<<grid, 256>>>(darrs, out, n);
                if (variant == 2) load_planar<NARR><<<grid, 256>>>(darrs, out, n);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
            }
            const char* nm[3] = {"strided-AoS", "lds-transposed-AoS", "planar"};
            printf("LOAD %-20s grid %6d  %8.3f ms  %8.1f GB/s\n", nm[variant], grid, best, bytes / best * 1e-6);
        }
    }
    {
        float best = 1e9; size_t n16 = n * 4;  // 256 MiB in, 256 MiB out
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipEventRecord(e0));
            copy16<<<8192, 256>>>(h[0], h[1], n16);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        printf("COPY dwordx4 256MiB->256MiB  %8.3f ms  %8.1f GB/s (read+write)\n", best, 2.0 * n16 * 16 / best * 1e-6);
    }
    return 0;
}
