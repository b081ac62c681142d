// mulrate29.hip -- would UNSATURATED 29-bit limbs beat the 32-bit-limb Montgomery block of the BN254 scalar-mul loop?
// The loop is VALU-issue-bound and 227 k of its 256 k non-multiplier instructions are the carry folds inside the Montgomery blocks (one add per
// v_mad_u64_u32: the instruction has a 64-bit addend but no carry-in).  With 9 limbs of 29 bits a product is below 2^58, so a whole column of
// a product-scanning multiplication (9 a*b terms + 9 m*p terms + the incoming carry < 2^62.2) accumulates in ONE 64-bit register pair through the
// addend of v_mad_u64_u32 with no carry handling at all; a column costs a shift and a mask.  171 multiplier instructions instead of 136, but ~60
// others instead of 162.  This probe measures a dependent chain of such multiplications (compiled C++: the inner statement IS one v_mad_u64_u32)
// against the 32-bit block of probes/mulrate.hip on the same harness, and checks one product against the value Python computes
// (tools/check_mul29.py reads the printed limbs).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I ark-mpc_amd/csrc probes/mulrate29.hip -o probes/mulrate29
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "fp_asm.hpp"
constexpr int FQ = F_BN254_FQ;
typedef unsigned long long u64_;
// BN254 Fq in 9 limbs of 29 bits, and -q^-1 mod 2^29
__device__ __constant__ u32 Q29[9] = {0x187cfd47u, 0x10460b6u, 0x1c72a34fu, 0x2d522d0u, 0x1585d978u, 0x2db40c0u, 0x0a6e141u, 0x0e5c2634u, 0x0030644eu};
constexpr u32 QINV29 = 0x04866389u;   // -q^-1 mod 2^29 (tools/check_mul29.py asserts it)
constexpr u32 M29 = (1u << 29) - 1;

struct F29 { u32 v[9]; };
__device__ __forceinline__ F29 mul29(const F29& a, const F29& b) {
    u64_ c = 0;
    u32 m[9];
    F29 o;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
#pragma unroll
        for (int i = 0; i <= k; ++i) c += (u64_)a.v[i] * b.v[k - i];
#pragma unroll
        for (int i = 0; i < k; ++i) c += (u64_)m[i] * Q29[k - i];
        m[k] = ((u32)c * QINV29) & M29;
        c += (u64_)m[k] * Q29[0];
        c >>= 29;
    }
#pragma unroll
    for (int k = 9; k < 17; ++k) {
#pragma unroll
        for (int i = k - 8; i <= 8; ++i) c += (u64_)a.v[i] * b.v[k - i];
#pragma unroll
        for (int i = k - 8; i <= 8; ++i) c += (u64_)m[i] * Q29[k - i];
        o.v[k - 9] = (u32)c & M29;
        c >>= 29;
    }
    o.v[8] = (u32)c;
    return o;
}
template <int MODE>   // 0: 29-bit compiled, 1: 32-bit hand-scheduled block
__global__ void __launch_bounds__(256) k_chain(const u32* in, u32* out, int reps) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (MODE == 0) {
        F29 x, y;
#pragma unroll
        for (int k = 0; k < 9; ++k) { y.v[k] = in[9 * i + k] & M29; x.v[k] = y.v[k]; }
        for (int r = 0; r < reps; ++r) x = mul29(x, y);
#pragma unroll
        for (int k = 0; k < 9; ++k) out[9 * i + k] = x.v[k];
    } else {
        Fe y, x;
#pragma unroll
        for (int k = 0; k < 8; ++k) { y.v[k] = in[9 * i + k] & (k == 7 ? 0x1fffffffu : 0xffffffffu); x.v[k] = y.v[k]; }
        for (int r = 0; r < reps; ++r) x = fe_mont_mul_asm<FQ>(x, y);
#pragma unroll
        for (int k = 0; k < 8; ++k) out[9 * i + k] = x.v[k];
    }
}
int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 2000;
    const size_t n = (size_t)256 * 256 * 16;
    u32 *in, *out;
    hipMalloc(&in, n * 36); hipMalloc(&out, n * 36);
    u32* h = (u32*)malloc(n * 36);
    for (size_t k = 0; k < 9 * n; ++k) h[k] = (u32)(0x9E3779B9u * (k + 1)) & ((k % 9 == 8) ? 0x1fffffu : M29);     // top limb small: value < 2^253 < q
    hipMemcpy(in, h, n * 36, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    // correctness: ONE multiplication x*x/2^261 mod q of element 0, printed for tools/check_mul29.py
    hipLaunchKernelGGL(k_chain<0>, dim3(1), dim3(256), 0, 0, in, out, 1);
    hipMemcpy(h + 9 * n - 9 * 256, out, 36, hipMemcpyDeviceToHost);
    printf("{\"check\": \"mul29\", \"x\": [");
    for (int k = 0; k < 9; ++k) printf("%u%s", h[k], k < 8 ? ", " : "");
    printf("], \"x_times_x\": [");
    for (int k = 0; k < 9; ++k) printf("%u%s", h[9 * n - 9 * 256 + k], k < 8 ? ", " : "");
    printf("]}\n");
    for (int mode = 0; mode < 2; ++mode) {
        auto kern = mode == 0 ? k_chain<0> : k_chain<1>;
        hipLaunchKernelGGL(kern, dim3(n / 256), dim3(256), 0, 0, in, out, 10);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(n / 256), dim3(256), 0, 0, in, out, reps);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double muls = (double)n * reps;
        printf("{\"block\": \"%s\", \"ms\": %.3f, \"fq_mul_per_s\": %.4e, \"cycles_per_wave_mul_at_2.4GHz\": %.0f}\n",
               mode == 0 ? "29-bit limbs, product scanning, compiled" : "32-bit limbs, hand-scheduled CIOS (fe_mont_mul_asm)", ms, muls / (ms * 1e-3),
               1024.0 * 2.4e9 / (muls / 64 / (ms * 1e-3)));
    }
    return 0;
}
