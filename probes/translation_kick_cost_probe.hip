// translation_kick_cost_probe.hip -- what the "translation kick" costs (map + unmap of one 4 KiB page of our own after registering a caller's
// vector in place, csrc/arkmpc_internal.hpp translation_kick): per call, idle and with a kernel running on the device.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/tkc probes/translation_kick_cost_probe.hip && /tmp/tkc
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
__global__ void spin(unsigned long long cycles, int* out) {
    const unsigned long long t0 = clock64();
    while (clock64() - t0 < cycles) {}
    if (out) *out = 1;
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    void* page = aligned_alloc(4096, 4096);
    const size_t big = 64u << 20;
    char* vec = (char*)aligned_alloc(4096, big);
    for (size_t i = 0; i < big; i += 4096) vec[i] = 1;
    (void)hipFree(0);
    ((char*)page)[0] = 1;
    for (int busy = 0; busy < 2; ++busy) {
        hipStream_t st; (void)hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
        if (busy) for (int k = 0; k < 4000; ++k) spin<<<256, 64, 0, st>>>(200000ull, nullptr);      // ~0.1 ms kernels back to back on another stream
        const int reps = 500;
        double t0 = now();
        int ok = 0; hipError_t last = hipSuccess;
        for (int i = 0; i < reps; ++i) { last = hipHostRegister(page, 4096, hipHostRegisterDefault); if (last == hipSuccess) { void* d = nullptr; (void)hipHostGetDevicePointer(&d, page, 0); hipHostUnregister(page); ++ok; } }
        const double kick_us = (now() - t0) / reps * 1e6;
        t0 = now();
        int ok2 = 0;
        for (int i = 0; i < 20; ++i) { if (hipHostRegister(vec, big, hipHostRegisterDefault) == hipSuccess) { void* d = nullptr; (void)hipHostGetDevicePointer(&d, vec, 0); ++ok2; hipHostUnregister(vec); } }
        const double big_ms = (now() - t0) / 20 * 1e3;
        printf("{\"device_busy\": %s, \"kick_us\": %.1f, \"kicks_ok\": %d, \"last_error\": \"%s\", \"register_plus_unregister_64MiB_ms\": %.3f, \"big_ok\": %d}\n", busy ? "true" : "false", kick_us, ok, hipGetErrorString(last), big_ms, ok2);
        hipDeviceSynchronize();
    }
    return 0;
}
