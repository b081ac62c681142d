// arkmpc_internal.hpp -- context, error plumbing and host-buffer staging shared by the C-ABI TUs.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>
#include <cstdio>
#include <cstring>
#include "../../include/arkmpc.h"
#include "fp.hpp"

struct arkmpc_ctx {
    int field_id = 0;
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    bool host_buffers = false;
    std::mutex mu;
    // last error text: written through ark_set_err only (its own lock: some entry points -- the batch carrier, the argument checks
    // before a CtxGuard -- report errors without holding `mu`, and arkmpc_last_error may run on another thread)
    std::mutex err_mu;
    std::string err;
    // device scratch arena for host-buffer staging and internal temporaries
    char* arena = nullptr;
    size_t arena_cap = 0;
    // small device word for reductions (mac_verify) + pinned mirror
    int* d_flag = nullptr;
    int* h_flag = nullptr;
    // MAC-verify flag: ONE word of host-coherent mapped memory that failing waves store into directly (no memset / D2H per call);
    // h_vflag is the host address, d_vflag its device alias.  Sticky between arkmpc_mac_verify_async calls.
    int* h_vflag = nullptr;
    int* d_vflag = nullptr;
    int* d_vgate = nullptr;     // device word: admits one host store per failed verification
    // pinned double buffer for the commitment pipeline
    unsigned char* h_pin[2] = {nullptr, nullptr};
    size_t h_pin_cap = 0;
    hipEvent_t ev[2] = {nullptr, nullptr};
    // small pinned buffer (64 KiB): window sums of the MSM for the host-side Horner fold
    unsigned char* h_small = nullptr;
    // per-context block cache in front of the device pool (arkmpc_malloc / arkmpc_free): a block freed through this context was last
    // used on THIS context's stream (the contract of arkmpc_free), so the same context may hand it out again with no event at all --
    // whatever uses it next is ordered behind the earlier use by the stream itself.  Keyed by size class; spills to the device pool
    // (with an event) above cache_cap.
    std::unordered_map<size_t, std::vector<void*>> cache;
    size_t cache_bytes = 0;
    // host link of the streaming host-to-host path (arkmpc_stream.inc): one copy stream per direction beside the compute stream, created on
    // first use, and a free list of timing-disabled events
    hipStream_t up = nullptr, down = nullptr;
    std::vector<hipEvent_t> link_ev;
    // kernel timer: event pairs bound to the dispatch of the NEXT K1 / K3 launch (hipExtLaunchKernelGGL)
    static constexpr int kTimerSlots = 64;
    hipEvent_t tev[2 * kTimerSlots] = {};
    int timer_slot = -1;
};

struct arkmpc_ctx;
static inline void ark_set_err(arkmpc_ctx* ctx, const std::string& what);
#define ARK_HIP(ctx, call)                                                                      \
    do {                                                                                        \
        hipError_t e__ = (call);                                                                \
        if (e__ != hipSuccess) {                                                                \
            ark_set_err((ctx), std::string(#call) + ": " + hipGetErrorString(e__));             \
            return ARKMPC_ERR_HIP;                                                              \
        }                                                                                       \
    } while (0)

static inline void ark_set_err(arkmpc_ctx* ctx, const std::string& what) {
    if (!ctx) return;
    std::lock_guard<std::mutex> lk(ctx->err_mu);
    ctx->err = what;
}
static inline int ark_bad(arkmpc_ctx* ctx, const char* what) {
    ark_set_err(ctx, what);
    return ARKMPC_ERR_BAD_ARG;
}

// RAII: serialise calls on the context and make its device current
struct CtxGuard {
    arkmpc_ctx* c;
    std::unique_lock<std::mutex> lk;
    int rc = ARKMPC_OK;
    explicit CtxGuard(arkmpc_ctx* ctx) : c(ctx), lk(ctx->mu) {
        hipError_t e = hipSetDevice(ctx->device);
        if (e != hipSuccess) { ark_set_err(ctx, std::string("hipSetDevice: ") + hipGetErrorString(e)); rc = ARKMPC_ERR_HIP; }
    }
};

// Arena: bump allocator over one device allocation, reset at the end of each staged call.
static inline int arena_reserve(arkmpc_ctx* ctx, size_t bytes) {
    if (bytes <= ctx->arena_cap) return ARKMPC_OK;
    ARK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->arena) ARK_HIP(ctx, hipFree(ctx->arena));
    ctx->arena = nullptr; ctx->arena_cap = 0;
    size_t cap = bytes + (bytes >> 2) + (1 << 20);
    ARK_HIP(ctx, hipMalloc((void**)&ctx->arena, cap));
    ctx->arena_cap = cap;
    return ARKMPC_OK;
}

// Staging of one API call.  In device mode `in`/`out` are pass-through (with an alignment check);
// in host mode they carve device buffers out of the arena, upload inputs and remember outputs.
struct Stage {
    arkmpc_ctx* ctx;
    int rc = ARKMPC_OK;
    struct In { const void* host; size_t bytes; size_t off; };
    std::vector<In> ins;
    struct OutPlan { void* host; size_t bytes; size_t off; };
    std::vector<OutPlan> oplan;
    size_t total = 0;
    explicit Stage(arkmpc_ctx* c) : ctx(c) {}

    static size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }

    // pass 1 (host mode): declare buffers; pass 2: resolve.  To keep call sites linear we do a
    // two-phase protocol: declare_* returns an index, then commit() uploads, and ptr(i) resolves.
    int declare_in(const void* p, size_t bytes) {
        if (!p && bytes) { rc = ark_bad(ctx, "null input pointer"); }
        ins.push_back({p, bytes, total}); total += align_up(bytes);
        return (int)ins.size() - 1;
    }
    int declare_out(void* p, size_t bytes) {
        if (!p && bytes) { rc = ark_bad(ctx, "null output pointer"); }
        oplan.push_back({p, bytes, total}); total += align_up(bytes);
        return (int)oplan.size() - 1;
    }
    // device-only scratch (both buffer modes): carved from the arena
    std::vector<size_t> scratch_off;
    size_t scratch_total = 0;
    int declare_scratch(size_t bytes) {
        scratch_off.push_back(scratch_total); scratch_total += align_up(bytes);
        return (int)scratch_off.size() - 1;
    }
    template <class T> T* scratch(int idx) const {
        return (T*)(ctx->arena + (ctx->host_buffers ? total : 0) + scratch_off[idx]);
    }
    int commit() {
        if (rc) return rc;
        if (scratch_total) {
            rc = arena_reserve(ctx, (ctx->host_buffers ? total : 0) + scratch_total);
            if (rc) return rc;
        }
        if (!ctx->host_buffers) {
            for (auto& i : ins) if (((uintptr_t)i.host & 15) != 0) return rc = ark_bad(ctx, "device pointer not 16-byte aligned");
            for (auto& o : oplan) if (((uintptr_t)o.host & 15) != 0) return rc = ark_bad(ctx, "device pointer not 16-byte aligned");
            return ARKMPC_OK;
        }
        rc = arena_reserve(ctx, total + scratch_total);
        if (rc) return rc;
        for (auto& i : ins) {
            if (!i.bytes) continue;
            hipError_t e = hipMemcpyAsync(ctx->arena + i.off, i.host, i.bytes, hipMemcpyHostToDevice, ctx->stream);
            if (e != hipSuccess) { ark_set_err(ctx, std::string("H2D: ") + hipGetErrorString(e)); return rc = ARKMPC_ERR_HIP; }
        }
        return ARKMPC_OK;
    }
    template <class T> const T* in(int idx) const {
        return ctx->host_buffers ? (const T*)(ctx->arena + ins[idx].off) : (const T*)ins[idx].host;
    }
    template <class T> T* out(int idx) const {
        return ctx->host_buffers ? (T*)(ctx->arena + oplan[idx].off) : (T*)oplan[idx].host;
    }
    // download outputs (host mode) and, in host mode, block until they have landed
    int finish() {
        if (rc) return rc;
        hipError_t le = hipGetLastError();
        if (le != hipSuccess) { ark_set_err(ctx, std::string("kernel launch: ") + hipGetErrorString(le)); return ARKMPC_ERR_HIP; }
        if (!ctx->host_buffers) return ARKMPC_OK;
        for (auto& o : oplan) {
            if (!o.bytes) continue;
            hipError_t e = hipMemcpyAsync(o.host, ctx->arena + o.off, o.bytes, hipMemcpyDeviceToHost, ctx->stream);
            if (e != hipSuccess) { ark_set_err(ctx, std::string("D2H: ") + hipGetErrorString(e)); return ARKMPC_ERR_HIP; }
        }
        hipError_t e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) { ark_set_err(ctx, std::string("sync: ") + hipGetErrorString(e)); return ARKMPC_ERR_HIP; }
        return ARKMPC_OK;
    }
};

// launch on the context's stream; if a kernel-timer slot is armed, bind its start/stop events to this dispatch
template <class K, class... A>
static inline void launch_k_lds(arkmpc_ctx* ctx, unsigned lds_bytes, K kernel, dim3 grid, dim3 block, A... args) {      // launch_k with `lds_bytes` of dynamic LDS
    if (ctx->timer_slot >= 0) {
        const int s = ctx->timer_slot;
        ctx->timer_slot = -1;
        hipExtLaunchKernelGGL(kernel, grid, block, lds_bytes, ctx->stream, ctx->tev[2 * s], ctx->tev[2 * s + 1], 0, args...);
    } else {
        hipLaunchKernelGGL(kernel, grid, block, lds_bytes, ctx->stream, args...);
    }
}
template <typename K, typename... A>
static inline void launch_k(arkmpc_ctx* ctx, K kernel, dim3 grid, dim3 block, A... args) {
    // experiment hook: ARKMPC_TEST_DYN_LDS=<bytes> of unused dynamic LDS per workgroup caps the workgroups a CU can hold (occupancy sensitivity
    // of the streaming kernels: tools/k3_occupancy_probe.sh); 0 / unset in production
    static const unsigned dyn_lds = getenv("ARKMPC_TEST_DYN_LDS") ? (unsigned)atoi(getenv("ARKMPC_TEST_DYN_LDS")) : 0u;
    if (ctx->timer_slot >= 0) {
        const int s = ctx->timer_slot;
        ctx->timer_slot = -1;
        hipExtLaunchKernelGGL(kernel, grid, block, dyn_lds, ctx->stream, ctx->tev[2 * s], ctx->tev[2 * s + 1], 0, args...);
    } else {
        hipLaunchKernelGGL(kernel, grid, block, dyn_lds, ctx->stream, args...);
    }
}

static inline unsigned blocks_for(size_t n, unsigned threads) { return (unsigned)((n + threads - 1) / threads); }

// Fe kernel argument from a host 4 x u64
static inline Fe fe_from_host(const uint64_t v[4]) {
    Fe r;
    for (int i = 0; i < 4; ++i) { r.v[2 * i] = (u32)v[i]; r.v[2 * i + 1] = (u32)(v[i] >> 32); }
    return r;
}

// host-side SHA3-256 (sha3_host.cpp)
struct Sha3State { uint64_t st[25]; unsigned char buf[136]; size_t fill; };
void sha3_256_init(Sha3State* s);
void sha3_256_update(Sha3State* s, const unsigned char* msg, size_t len);
void sha3_256_final(Sha3State* s, unsigned char out[32]);
// integer value of 32 big-endian bytes reduced mod p, returned in Montgomery form (host big-int)
void host_from_be_bytes_mod_order(int field_id, const unsigned char be[32], uint64_t out_mont[4]);
void host_to_bytes_be(int field_id, const uint64_t mont[4], unsigned char out[32]);
