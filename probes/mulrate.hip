// mulrate.hip -- how fast does the hand-scheduled Montgomery multiplication block (fe_mont_mul_asm, BN254 Fq, lazy [0,2q) range)
// issue as a function of waves per SIMD?  Chains of dependent multiplications (1 or 2 independent chains per thread), occupancy
// capped through the dynamic-LDS request of a 256-thread workgroup (one wave per SIMD per workgroup).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I ark-mpc_amd/csrc probes/mulrate.hip -o probes/mulrate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "fp_asm.hpp"
constexpr int FQ = F_BN254_FQ;
template <int CH>
__global__ void __launch_bounds__(256) k_chain(const u64* in, u64* out, int reps) {
    extern __shared__ char lds[];
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    Fe x[CH], y = fe_load(in + 4 * i);
#pragma unroll
    for (int c = 0; c < CH; ++c) { x[c] = y; x[c].v[0] ^= c; }
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int c = 0; c < CH; ++c) x[c] = fe_mont_mul_asm<FQ>(x[c], y);
    }
    Fe s = x[0];
#pragma unroll
    for (int c = 1; c < CH; ++c) s.v[0] ^= x[c].v[3];
    if (reps < 0) lds[threadIdx.x] = 1;
    fe_store(out + 4 * i, s);
}
int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 2000;
    const size_t n = (size_t)256 * 256 * 16;     // 4096 workgroups of 256 threads
    u64 *in, *out;
    hipMalloc(&in, n * 32); hipMalloc(&out, n * 32);
    hipMemset(in, 0x11, n * 32);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int ch = 1; ch <= 2; ++ch)
        for (int w : {1, 2, 3, 4, 5, 8}) {
            const size_t lds = (160 * 1024) / w - 512;      // w workgroups (= w waves per SIMD) fit per CU
            auto kern = ch == 1 ? k_chain<1> : k_chain<2>;
            hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL(kern, dim3(n / 256), dim3(256), lds, 0, in, out, 10);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(kern, dim3(n / 256), dim3(256), lds, 0, in, out, reps);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double muls = (double)n * reps * ch;
            printf("{\"chains\": %d, \"waves_per_simd\": %d, \"ms\": %.3f, \"fq_mul_per_s\": %.4e, \"frac_of_mad_peak\": %.3f}\n", ch, w, ms, muls / (ms * 1e-3),
                   muls / (ms * 1e-3) * 136 / 31.2e12);
        }
    return 0;
}
