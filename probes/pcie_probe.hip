// pcie_probe.hip -- what the host link of this box gives the streaming host-to-host path (csrc/arkmpc_stream.hip), measured before
// designing it: (1) cost of hipHostRegister on a caller's pageable buffer, (2) DMA rates from registered / pinned memory, one direction
// and both, by chunk size, (3) what hipMemcpyAsync does with PAGEABLE memory on two streams from two host threads, (4) the rate at which
// T host threads can memcpy into / out of a pinned staging ring, (5) a kernel reading / writing registered host memory directly.
//   hipcc --offload-arch=gfx950 -O3 -pthread -o probes/pcie_probe probes/pcie_probe.hip && probes/pcie_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

__global__ void k_copy(const uint4* __restrict__ in, uint4* __restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i];
}

// the share halves (first 32 B) of 64-byte records in host memory -> packed 32-byte elements in device memory: two lanes per record
__global__ void k_gather_halves(const uint4* __restrict__ in, uint4* __restrict__ out, size_t nrec) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < 2 * nrec; i += (size_t)gridDim.x * blockDim.x) out[i] = in[4 * (i >> 1) + (i & 1)];
}

static void par_memcpy(char* dst, const char* src, size_t bytes, int T) {
    std::vector<std::thread> th;
    size_t per = (bytes / T + 4095) & ~(size_t)4095;
    for (int t = 0; t < T; ++t) {
        size_t lo = (size_t)t * per; if (lo >= bytes) break;
        size_t cnt = bytes - lo < per ? bytes - lo : per;
        th.emplace_back([=] { memcpy(dst + lo, src + lo, cnt); });
    }
    for (auto& t : th) t.join();
}

int main() {
    CK(hipSetDevice(0));
    const size_t MB = 1 << 20;
    const size_t big = 256 * MB;
    char *d0, *d1;
    CK(hipMalloc((void**)&d0, big)); CK(hipMalloc((void**)&d1, big));
    hipStream_t s0, s1, s2;
    CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));

    // (1) hipHostRegister cost
    for (size_t sz : {16 * MB, 64 * MB, 256 * MB}) {
        char* p = (char*)aligned_alloc(4096, sz);
        memset(p, 1, sz);
        double t0 = now();
        CK(hipHostRegister(p, sz, hipHostRegisterDefault));
        double t1 = now();
        CK(hipHostUnregister(p));
        double t2 = now();
        printf("{\"probe\":\"host_register\",\"MiB\":%zu,\"register_ms\":%.3f,\"register_GBps\":%.2f,\"unregister_ms\":%.3f}\n", sz / MB, (t1 - t0) * 1e3, sz / (t1 - t0) / 1e9, (t2 - t1) * 1e3);
        free(p);
    }
    // (2) DMA from pinned (hipHostMalloc) and registered memory by chunk size
    char *h0, *h1;
    CK(hipHostMalloc((void**)&h0, big, hipHostMallocDefault)); CK(hipHostMalloc((void**)&h1, big, hipHostMallocDefault));
    memset(h0, 2, big); memset(h1, 3, big);
    char* r0 = (char*)aligned_alloc(4096, big); char* r1 = (char*)aligned_alloc(4096, big);
    memset(r0, 4, big); memset(r1, 5, big);
    CK(hipHostRegister(r0, big, hipHostRegisterDefault)); CK(hipHostRegister(r1, big, hipHostRegisterDefault));
    for (int kind = 0; kind < 2; ++kind) {
        char* a = kind ? r0 : h0; char* b = kind ? r1 : h1;
        for (size_t ch : {1 * MB, 4 * MB, 16 * MB, 64 * MB}) {
            const int reps = 3;
            auto run = [&](int mode) {
                CK(hipDeviceSynchronize());
                double t0 = now();
                for (int r = 0; r < reps; ++r)
                    for (size_t off = 0; off < big; off += ch) {
                        if (mode != 1) CK(hipMemcpyAsync(d0 + off, a + off, ch, hipMemcpyHostToDevice, s0));
                        if (mode != 0) CK(hipMemcpyAsync(b + off, d1 + off, ch, hipMemcpyDeviceToHost, s1));
                    }
                CK(hipDeviceSynchronize());
                return (now() - t0) / reps;
            };
            double th = run(0), td = run(1), tb = run(2);
            printf("{\"probe\":\"dma\",\"memory\":\"%s\",\"chunk_MiB\":%zu,\"h2d_GBps\":%.1f,\"d2h_GBps\":%.1f,\"both_total_GBps\":%.1f,\"both_h2d_GBps\":%.1f}\n", kind ? "registered" : "hipHostMalloc",
                   ch / MB, big / th / 1e9, big / td / 1e9, 2 * big / tb / 1e9, big / tb / 1e9);
        }
    }
    // asymmetric duplex: 3 parts up, 1 part down (the path's 384 B up / 128 B down per gate), 8 MiB chunks
    {
        const size_t ch = 8 * MB;
        CK(hipDeviceSynchronize());
        double t0 = now();
        size_t up = 0, down = 0;
        for (size_t off = 0; off < big; off += ch) {
            CK(hipMemcpyAsync(d0 + off, h0 + off, ch, hipMemcpyHostToDevice, s0)); up += ch;
            if ((off / ch) % 3 == 2) { CK(hipMemcpyAsync(h1 + down, d1 + down, ch, hipMemcpyDeviceToHost, s1)); down += ch; }
        }
        CK(hipDeviceSynchronize());
        double t = now() - t0;
        printf("{\"probe\":\"dma_3up_1down\",\"h2d_GBps\":%.1f,\"d2h_GBps\":%.1f}\n", up / t / 1e9, down / t / 1e9);
    }
    // (3) pageable memory through hipMemcpyAsync: one thread, and two threads on two streams (does the runtime overlap them?)
    {
        char* p0 = (char*)aligned_alloc(4096, big); char* p1 = (char*)aligned_alloc(4096, big);
        memset(p0, 6, big); memset(p1, 7, big);
        for (size_t ch : {4 * MB, 16 * MB, 256 * MB}) {
            CK(hipDeviceSynchronize());
            double t0 = now();
            for (size_t off = 0; off < big; off += ch) CK(hipMemcpyAsync(d0 + off, p0 + off, ch, hipMemcpyHostToDevice, s0));
            double tcall = now() - t0;
            CK(hipDeviceSynchronize());
            double th = now() - t0;
            t0 = now();
            for (size_t off = 0; off < big; off += ch) CK(hipMemcpyAsync(p1 + off, d1 + off, ch, hipMemcpyDeviceToHost, s1));
            CK(hipDeviceSynchronize());
            double td = now() - t0;
            t0 = now();
            std::thread ta([&] { CK(hipSetDevice(0)); for (size_t off = 0; off < big; off += ch) CK(hipMemcpyAsync(d0 + off, p0 + off, ch, hipMemcpyHostToDevice, s0)); CK(hipStreamSynchronize(s0)); });
            std::thread tb([&] { CK(hipSetDevice(0)); for (size_t off = 0; off < big; off += ch) CK(hipMemcpyAsync(p1 + off, d1 + off, ch, hipMemcpyDeviceToHost, s1)); CK(hipStreamSynchronize(s1)); });
            ta.join(); tb.join();
            double tb2 = now() - t0;
            printf("{\"probe\":\"pageable\",\"chunk_MiB\":%zu,\"h2d_GBps\":%.1f,\"h2d_call_returns_after_frac\":%.2f,\"d2h_GBps\":%.1f,\"two_threads_both_total_GBps\":%.1f}\n", ch / MB, big / th / 1e9,
                   tcall / th, big / td / 1e9, 2 * big / tb2 / 1e9);
        }
        // (4) host threads copying pageable -> pinned ring (and back): the staging rate
        for (int T : {1, 2, 4, 8, 16, 32}) {
            double t0 = now();
            par_memcpy(h0, p0, big, T);
            double t_in = now() - t0;
            t0 = now();
            par_memcpy(p1, h1, big, T);
            double t_out = now() - t0;
            printf("{\"probe\":\"staging_memcpy\",\"threads\":%d,\"pageable_to_pinned_GBps\":%.1f,\"pinned_to_pageable_GBps\":%.1f}\n", T, big / t_in / 1e9, big / t_out / 1e9);
        }
        free(p0); free(p1);
    }
    // (5) kernels addressing registered host memory directly (zero-copy): read host -> write device, read device -> write host, both at once
    {
        char *dr0, *dr1;
        CK(hipHostGetDevicePointer((void**)&dr0, h0, 0)); CK(hipHostGetDevicePointer((void**)&dr1, h1, 0));
        const size_t nv = big / 16;
        for (int blocks : {256, 1024, 4096}) {
            auto t = [&](int mode) {
                CK(hipDeviceSynchronize());
                double t0 = now();
                for (int r = 0; r < 3; ++r) {
                    if (mode != 1) hipLaunchKernelGGL(k_copy, dim3(blocks), dim3(256), 0, s0, (const uint4*)dr0, (uint4*)d0, nv);
                    if (mode != 0) hipLaunchKernelGGL(k_copy, dim3(blocks), dim3(256), 0, s1, (const uint4*)d1, (uint4*)dr1, nv);
                }
                CK(hipDeviceSynchronize());
                return (now() - t0) / 3;
            };
            double tr = t(0), tw = t(1), tb = t(2);
            printf("{\"probe\":\"zero_copy_kernel\",\"blocks\":%d,\"read_host_GBps\":%.1f,\"write_host_GBps\":%.1f,\"both_total_GBps\":%.1f}\n", blocks, big / tr / 1e9, big / tw / 1e9, 2 * big / tb / 1e9);
        }
    }
    // (6) only the share halves of ScalarShare records cross the link: a kernel gathering them from mapped host memory, and the DMA engines' 2D copy
    {
        char* dr0; CK(hipHostGetDevicePointer((void**)&dr0, h0, 0));
        const size_t nrec = big / 64;
        for (int blocks : {256, 1024, 4096}) {
            CK(hipDeviceSynchronize());
            double t0 = now();
            for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k_gather_halves, dim3(blocks), dim3(256), 0, s0, (const uint4*)dr0, (uint4*)d0, nrec);
            CK(hipDeviceSynchronize());
            double t = (now() - t0) / 3;
            printf("{\"probe\":\"gather_share_halves_kernel\",\"blocks\":%d,\"useful_GBps\":%.1f,\"span_GBps\":%.1f}\n", blocks, nrec * 32 / t / 1e9, nrec * 64 / t / 1e9);
        }
        CK(hipDeviceSynchronize());
        double t0 = now();
        CK(hipMemcpy2DAsync(d0, 32, h0, 64, 32, nrec, hipMemcpyHostToDevice, s0));
        CK(hipDeviceSynchronize());
        double t = now() - t0;
        printf("{\"probe\":\"memcpy2d_width32_pitch64\",\"useful_GBps\":%.1f}\n", nrec * 32 / t / 1e9);
    }
    unsigned nthreads = std::thread::hardware_concurrency();
    printf("{\"probe\":\"host\",\"hardware_threads\":%u}\n", nthreads);
    return 0;
}
