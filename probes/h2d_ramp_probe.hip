// h2d_ramp_probe.hip -- per-copy duration of six back-to-back 64 MiB pinned H2D copies on one stream, repeated with and without an idle gap
// before each burst, and with a concurrent D2H stream: does the link start slow after idling?  (Seen in the copy trace of the streaming
// sessions: the first ~128 MiB of every session go up at ~43 GB/s, the rest at 56-57.)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
int main() {
    const size_t sz = 64u << 20; const int N = 6;
    char *h, *d, *h2, *d2;
    CK(hipHostMalloc((void**)&h, sz * N, 0)); CK(hipMalloc((void**)&d, sz * N)); memset(h, 1, sz * N);
    CK(hipHostMalloc((void**)&h2, sz, 0)); CK(hipMalloc((void**)&d2, sz));
    hipStream_t s, s2; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    hipEvent_t ev[N + 1]; for (auto& e : ev) CK(hipEventCreate(&e));
    for (int gap_ms : {0, 1, 5, 50}) {
        for (int it = 0; it < 3; ++it) {
            std::this_thread::sleep_for(std::chrono::milliseconds(gap_ms));
            CK(hipEventRecord(ev[0], s));
            for (int k = 0; k < N; ++k) { CK(hipMemcpyAsync(d + k * sz, h + k * sz, sz, hipMemcpyHostToDevice, s)); CK(hipEventRecord(ev[k + 1], s)); }
            CK(hipStreamSynchronize(s));
            printf("{\"idle_ms_before\":%d,\"per_copy_GBps\":[", gap_ms);
            for (int k = 0; k < N; ++k) { float ms; CK(hipEventElapsedTime(&ms, ev[k], ev[k + 1])); printf("%s%.1f", k ? "," : "", sz / (ms * 1e-3) / 1e9); }
            printf("]}\n");
        }
    }
    return 0;
}
