// zero_copy_session_probe.hip -- would a streaming session made of kernels that address the caller's pinned vectors DIRECTLY (no copy
// commands at all) beat the three-stream DMA pipeline of csrc/arkmpc_stream.inc (8.0-8.1 ms per 2^20 gates)?  The DMA session loses
// ~0.4 ms to the 17-30 us between consecutive copies of a stream and ~0.5 ms to downloads running flat out beside the uploads; kernels
// have no gaps to speak of and pace their writes by their reads.  Stand-in arithmetic (xor) with the session's exact traffic:
//   phase 1: read x, y, a, b records from HOST (256 B per gate), write d||e to HOST (64 B) and to HBM, stash a, b in HBM;
//   phase 2: read c and the peer's d||e from HOST (128 B), a, b, own d||e from HBM, write the result record to HOST (64 B).
//   hipcc --offload-arch=gfx950 -O3 -o probes/zero_copy_session_probe probes/zero_copy_session_probe.hip && probes/zero_copy_session_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

__device__ inline uint4 x4(uint4 a, uint4 b) { return make_uint4(a.x ^ b.x, a.y ^ b.y, a.z ^ b.z, a.w ^ b.w); }

// one thread per 16-byte quarter of a gate's records; gates [lo, lo + cnt)
__global__ void k1z(size_t lo, size_t cnt, size_t n, const uint4* __restrict__ x, const uint4* __restrict__ y, const uint4* __restrict__ a, const uint4* __restrict__ b,
                    uint4* __restrict__ de_host, uint4* __restrict__ de_dev, uint4* __restrict__ a_dev, uint4* __restrict__ b_dev) {
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < 4 * cnt; t += (size_t)gridDim.x * blockDim.x) {
        const size_t g = lo + (t >> 2), q = t & 3, i = 4 * g + q;
        const uint4 vx = x[i], vy = y[i], va = a[i], vb = b[i];
        a_dev[i] = va; b_dev[i] = vb;
        if (q < 2) {
            const uint4 d = x4(vx, va), e = x4(vy, vb);
            de_host[2 * g + q] = d; de_host[2 * (n + g) + q] = e;
            de_dev[2 * g + q] = d; de_dev[2 * (n + g) + q] = e;
        }
    }
}
__global__ void k3z(size_t lo, size_t cnt, size_t n, const uint4* __restrict__ c, const uint4* __restrict__ peer, const uint4* __restrict__ de_dev,
                    const uint4* __restrict__ a_dev, const uint4* __restrict__ b_dev, uint4* __restrict__ out) {
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < 4 * cnt; t += (size_t)gridDim.x * blockDim.x) {
        const size_t g = lo + (t >> 2), q = t & 3, i = 4 * g + q, h = q & 1;
        const uint4 vc = c[i], pd = peer[2 * g + h], pe = peer[2 * (n + g) + h];
        const uint4 d = x4(de_dev[2 * g + h], pd), e = x4(de_dev[2 * (n + g) + h], pe);
        out[i] = x4(x4(vc, x4(d, e)), x4(a_dev[i], b_dev[i]));
    }
}

int main(int argc, char** argv) {
    CK(hipSetDevice(0));
    const int lg = argc > 1 ? atoi(argv[1]) : 20;
    const size_t n = (size_t)1 << lg, rec = n * 64;
    uint4 *x, *y, *a, *b, *c, *peer, *de, *out;
    for (uint4** p : {&x, &y, &a, &b, &c, &peer, &de, &out}) { CK(hipHostMalloc((void**)p, rec, hipHostMallocDefault)); }
    unsigned s = 12345;
    for (uint4* p : {x, y, a, b, c, peer}) { unsigned* w = (unsigned*)p; for (size_t i = 0; i < rec / 4; ++i) { s = s * 1664525u + 1013904223u; w[i] = s; } }
    uint4 *dde, *da, *db, *dx, *dy, *dc, *dpeer, *dout;
    for (uint4** p : {&dde, &da, &db, &dx, &dy, &dc, &dpeer, &dout}) CK(hipMalloc((void**)p, rec));
    hipStream_t st, up, down;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&up, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&down, hipStreamNonBlocking));

    // reference points on this box, this run: bare upload of the session's 384 B per gate, one copy per array
    {
        std::vector<double> ts;
        for (int r = 0; r < 5; ++r) {
            CK(hipDeviceSynchronize());
            const double t0 = now();
            CK(hipMemcpyAsync(dx, x, rec, hipMemcpyHostToDevice, up)); CK(hipMemcpyAsync(dy, y, rec, hipMemcpyHostToDevice, up));
            CK(hipMemcpyAsync(da, a, rec, hipMemcpyHostToDevice, up)); CK(hipMemcpyAsync(db, b, rec, hipMemcpyHostToDevice, up));
            CK(hipMemcpyAsync(dc, c, rec, hipMemcpyHostToDevice, up)); CK(hipMemcpyAsync(dpeer, peer, rec, hipMemcpyHostToDevice, up));
            CK(hipDeviceSynchronize());
            ts.push_back(now() - t0);
        }
        std::sort(ts.begin(), ts.end());
        printf("{\"probe\":\"bare_upload_6_arrays\",\"log2n\":%d,\"ms\":%.3f,\"GBps\":%.1f}\n", lg, ts[2] * 1e3, 6 * rec / ts[2] / 1e9);
    }
    for (int blocks : {128, 256, 512, 1024, 2048}) {
        for (int chunks : {1, 4, 8}) {
            std::vector<double> ts, t1s;
            for (int r = 0; r < 6; ++r) {
                memset(de, 0, 4096); memset(out, 0, 4096);
                CK(hipDeviceSynchronize());
                const double t0 = now();
                const size_t per = n / chunks;
                for (int k = 0; k < chunks; ++k) hipLaunchKernelGGL(k1z, dim3(blocks), dim3(256), 0, st, k * per, per, n, x, y, a, b, de, dde, da, db);
                CK(hipStreamSynchronize(st));
                const double t1 = now();
                for (int k = 0; k < chunks; ++k) hipLaunchKernelGGL(k3z, dim3(blocks), dim3(256), 0, st, k * per, per, n, c, peer, dde, da, db, out);
                CK(hipStreamSynchronize(st));
                const double t2 = now();
                if (r) { ts.push_back(t2 - t0); t1s.push_back(t1 - t0); }
            }
            std::sort(ts.begin(), ts.end()); std::sort(t1s.begin(), t1s.end());
            // check a few gates
            bool ok = true;
            for (size_t g : {(size_t)0, n / 3, n - 1}) for (int q = 0; q < 2; ++q) {
                const uint4 d = de[2 * g + q], xx = x[4 * g + q], aa = a[4 * g + q];
                ok = ok && d.x == (xx.x ^ aa.x) && d.w == (xx.w ^ aa.w);
            }
            printf("{\"probe\":\"zero_copy_session\",\"log2n\":%d,\"blocks\":%d,\"chunks\":%d,\"session_ms\":%.3f,\"phase1_ms\":%.3f,\"phase2_ms\":%.3f,\"party_gates_per_s\":%.3e,\"up_GBps\":%.1f,\"ok\":%s}\n",
                   lg, blocks, chunks, ts[2] * 1e3, t1s[2] * 1e3, (ts[2] - t1s[2]) * 1e3, n / ts[2], n * 384 / ts[2] / 1e9, ok ? "true" : "false");
        }
    }
    // phase 2 overlapped with the previous... (both phases of two different sessions in flight: phase 2 of A on one stream beside phase 1 of B)
    for (int blocks : {256, 512}) {
        std::vector<double> ts;
        for (int r = 0; r < 5; ++r) {
            CK(hipDeviceSynchronize());
            const double t0 = now();
            hipLaunchKernelGGL(k1z, dim3(blocks), dim3(256), 0, st, 0, n, n, x, y, a, b, de, dde, da, db);
            hipLaunchKernelGGL(k3z, dim3(blocks), dim3(256), 0, up, 0, n, n, c, peer, dde, da, db, out);
            CK(hipDeviceSynchronize());
            ts.push_back(now() - t0);
        }
        std::sort(ts.begin(), ts.end());
        printf("{\"probe\":\"zero_copy_two_phases_concurrent\",\"blocks\":%d,\"ms\":%.3f,\"up_GBps\":%.1f}\n", blocks, ts[2] * 1e3, n * 384 / ts[2] / 1e9);
    }
    return 0;
}
