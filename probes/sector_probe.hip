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

// K1's own pattern on split columns: one 32-byte element per lane from each of four columns (two 16-byte loads 16 B apart, lanes 32 B apart),
// two 32-byte elements written.  LD: 0 plain, 1 non-temporal, 2 buffer loads with sc0 sc1 nt; ST: 0 plain, 1 non-temporal
template <int LD, int ST> __global__ void __launch_bounds__(256) k_k1like(size_t n, const unsigned char* x, const unsigned char* y, const unsigned char* a, const unsigned char* b,
                                                                         unsigned char* d, unsigned char* e) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    u32x4 v[8];
    const unsigned char* src[4] = {x, a, y, b};
    for (int k = 0; k < 4; ++k) {
        if (LD == 2) {
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src[k], 0, 0x7fffffff, 0x00027000);
            v[2 * k] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(32 * i), 0, 7);
            v[2 * k + 1] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(32 * i + 16), 0, 7);
        } else {
            const u32x4* p = (const u32x4*)(src[k] + 32 * i);
            v[2 * k] = LD == 1 ? __builtin_nontemporal_load(p) : p[0];
            v[2 * k + 1] = LD == 1 ? __builtin_nontemporal_load(p + 1) : p[1];
        }
    }
    const u32x4 d0 = v[0] - v[2], d1 = v[1] - v[3], e0 = v[4] - v[6], e1 = v[5] - v[7];
    u32x4* pd = (u32x4*)(d + 32 * i); u32x4* pe = (u32x4*)(e + 32 * i);
    if (ST == 1) { __builtin_nontemporal_store(d0, pd); __builtin_nontemporal_store(d1, pd + 1); __builtin_nontemporal_store(e0, pe); __builtin_nontemporal_store(e1, pe + 1); }
    else { pd[0] = d0; pd[1] = d1; pe[0] = e0; pe[1] = e1; }
}
#define K1_SETS 6
template <int LD, int ST> static void run_k1(const char* name, unsigned char* buf, size_t n) {
    // SETS rotating sets of (x, y, a, b, d, e), 32 n bytes each (inputs alone: 128 MiB per set), so that nothing of a launch is still in the
    // 256 MiB Infinity Cache when its set comes round again -- with two sets and non-temporal stores it is: 28.6 us, an artefact
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto launch = [&](int set) {
        unsigned char* base = buf + (size_t)set * 6 * 32 * n;
        hipLaunchKernelGGL((k_k1like<LD, ST>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, n, base, base + 32 * n, base + 64 * n, base + 96 * n, base + 128 * n, base + 160 * n);
    };
    for (int w = 0; w < K1_SETS; ++w) launch(w);
    CK(hipEventRecord(e0));
    const int reps = 48;
    for (int r = 0; r < reps; ++r) launch(r % K1_SETS);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    printf("{\"variant\": \"k1like_%s\", \"gates\": %zu, \"us\": %.2f, \"GBps_on_192B_per_gate\": %.1f}\n", name, n, ms * 1e3, n * 192.0 / (ms * 1e-3) / 1e9);
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
    CK(hipFree(buf));
    const size_t n = (size_t)1 << 20;             // K1 at the headline's batch: K1_SETS sets x 6 columns x 32 MiB
    unsigned char* cols; CK(hipMalloc(&cols, (size_t)K1_SETS * 6 * 32 * n)); CK(hipMemset(cols, 3, (size_t)K1_SETS * 6 * 32 * n));
    run_k1<0, 0>("plain_plain", cols, n);
    run_k1<1, 0>("nt_plain", cols, n);
    run_k1<1, 1>("nt_nt", cols, n);
    run_k1<2, 0>("buffer_sc0sc1nt_plain", cols, n);
    run_k1<2, 1>("buffer_sc0sc1nt_nt", cols, n);
    run_k1<0, 1>("plain_nt", cols, n);
    return 0;
}
