// nullstream_wait_probe.hip -- does hipStreamWaitEvent(NULL stream, event recorded on a non-blocking stream) order a kernel on the null stream
// behind a DMA on that stream?  (tests/test_gpu_stream.py: with the context bound to torch's default stream -- the legacy null stream -- K1
// of the first chunk ran before its chunk of b had landed.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
__global__ void k_check(const unsigned* p, size_t n, unsigned want, unsigned* bad) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && p[i] != want) atomicAdd(bad, 1u);
}
int main() {
    const size_t n = 16u << 20;  // 64 MiB of u32
    unsigned *h, *d, *bad, *hbad;
    CK(hipHostMalloc((void**)&h, n * 4, 0)); CK(hipMalloc((void**)&d, n * 4)); CK(hipMalloc((void**)&bad, 4)); CK(hipHostMalloc((void**)&hbad, 4, 0));
    hipStream_t up, other[6];
    CK(hipStreamCreateWithFlags(&up, hipStreamNonBlocking));
    for (auto& s : other) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t ev; CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    for (int mode = 0; mode < 2; ++mode) {            // 0: kernel on the NULL stream, 1: kernel on a non-blocking stream
        int fails = 0;
        for (unsigned it = 1; it <= 20; ++it) {
            for (size_t i = 0; i < n; ++i) h[i] = it;
            CK(hipMemsetAsync(bad, 0, 4, mode ? other[0] : nullptr));
            CK(hipStreamSynchronize(mode ? other[0] : nullptr));
            CK(hipMemcpyAsync(d, h, n * 4, hipMemcpyHostToDevice, up));
            CK(hipEventRecord(ev, up));
            hipStream_t cs = mode ? other[0] : nullptr;
            CK(hipStreamWaitEvent(cs, ev, 0));
            hipLaunchKernelGGL(k_check, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, cs, d, n, it, bad);
            CK(hipMemcpyAsync(hbad, bad, 4, hipMemcpyDeviceToHost, cs));
            CK(hipStreamSynchronize(cs));
            CK(hipStreamSynchronize(up));
            if (*hbad) ++fails;
        }
        printf("{\"kernel_stream\":\"%s\",\"iterations\":20,\"iterations_where_the_kernel_saw_stale_data\":%d}\n", mode ? "non-blocking" : "NULL", fails);
    }
    return 0;
}
