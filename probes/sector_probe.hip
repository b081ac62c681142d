// sector_probe.hip -- round 6: can a kernel that needs only the first 32 bytes of every 64-byte record (K1 on arkworks AoS records: the share
// halves of x, y, a, b) make HBM deliver only those 32 bytes?  K1 AoS moves 320.9 B per party-gate against 192 algorithmic (PMC), i.e. whole
// records.  This probe times the access pattern alone, with every load flavour the ISA offers, over 512 MiB (twice the Infinity Cache):
//   full      every byte of every record (4 lanes x 16 B per record): the reference rate
//   half2     2 lanes per record, 16 B each, bytes [0, 32) of each record          (what K1 AoS does)
//   half1     1 lane per record, two 16 B loads (offsets 0 and 16)
//   half2_nt  half2 with non-temporal loads;  half2_glc / _slc / _both: buffer loads with the cache-policy bits
//   half8     8-byte loads, 4 lanes per record on bytes [0, 32)
// prints one JSON line per variant: time per pass, "useful" GB/s (bytes asked for) and what that means if whole records were fetched
//   hipcc --offload-arch=gfx950 -O3 -o probes/sector_probe probes/sector_probe.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e__)); exit(2); } } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

template <int V> __global__ void __launch_bounds__(256) k_read(const unsigned char* __restrict__ base, size_t nrec, uint32_t* __restrict__ sink) {
    uint32_t acc = 0;
    const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x, nth = (size_t)gridDim.x * 256;
    if (V == 0) {           // full: lane -> 16-byte quarter q of the buffer
        for (size_t q = tid; q < 4 * nrec; q += nth) { const u32x4 v = *(const u32x4*)(base + 16 * q); acc += v.x ^ v.y ^ v.z ^ v.w; }
    } else if (V == 1 || V == 3) {    // half2 (plain / non-temporal)
        for (size_t h = tid; h < 2 * nrec; h += nth) {
            const u32x4* p = (const u32x4*)(base + 64 * (h >> 1) + 16 * (h & 1));
            const u32x4 v = V == 3 ? __builtin_nontemporal_load(p) : *p;
            acc += v.x ^ v.y ^ v.z ^ v.w;
        }
    } else if (V == 2) {    // half1
        for (size_t r = tid; r < nrec; r += nth) {
            const u32x4 v = *(const u32x4*)(base + 64 * r), w = *(const u32x4*)(base + 64 * r + 16);
            acc += v.x ^ v.y ^ v.z ^ v.w ^ w.x ^ w.y ^ w.z ^ w.w;
        }
    } else if (V == 7) {    // half8: 8-byte loads, 4 lanes per record
        for (size_t h = tid; h < 4 * nrec; h += nth) { const u32x2 v = *(const u32x2*)(base + 64 * (h >> 2) + 8 * (h & 3)); acc += v.x ^ v.y; }
    } else {                // buffer loads with cache-policy bits: 4 = glc (sc0), 5 = slc (nt), 6 = glc | slc | dlc (sc1)
        // one descriptor over the first 2 GiB is enough for 512 MiB
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7fffffff, 0x00027000);
        for (size_t h = tid; h < 2 * nrec; h += nth) {
            const int off = (int)(64 * (h >> 1) + 16 * (h & 1));
            u32x4 v;
            if (V == 4) v = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 1);
            else if (V == 5) v = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 2);
            else v = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 7);
            acc += v.x ^ v.y ^ v.z ^ v.w;
        }
    }
    if (acc == 0x12345678u) sink[tid & 1023] = acc;
}

template <int V> static void run(const char* name, const unsigned char* buf, size_t nrec, uint32_t* sink, double useful_per_rec) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k_read<V>, dim3(4096), dim3(256), 0, 0, buf, nrec, sink);
    CK(hipEventRecord(e0));
    const int reps = 10;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_read<V>, dim3(4096), dim3(256), 0, 0, buf, nrec, sink);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    printf("{\"variant\": \"%s\", \"ms\": %.4f, \"useful_GBps\": %.1f, \"GBps_if_whole_records_were_fetched\": %.1f}\n", name, ms, nrec * useful_per_rec / (ms * 1e-3) / 1e9,
           nrec * 64.0 / (ms * 1e-3) / 1e9);
}

int main() {
    const size_t nrec = (size_t)1 << 23;          // 2^23 records x 64 B = 512 MiB
    unsigned char* buf; uint32_t* sink;
    CK(hipMalloc(&buf, nrec * 64)); CK(hipMalloc(&sink, 4096)); CK(hipMemset(buf, 1, nrec * 64));
    run<0>("full", buf, nrec, sink, 64);
    run<1>("half2", buf, nrec, sink, 32);
    run<2>("half1", buf, nrec, sink, 32);
    run<3>("half2_nt", buf, nrec, sink, 32);
    run<4>("half2_buffer_glc", buf, nrec, sink, 32);
    run<5>("half2_buffer_slc", buf, nrec, sink, 32);
    run<6>("half2_buffer_glc_slc_dlc", buf, nrec, sink, 32);
    run<7>("half8", buf, nrec, sink, 32);
    return 0;
}
