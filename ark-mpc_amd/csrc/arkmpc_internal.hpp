// arkmpc_internal.hpp -- context, error plumbing and host-buffer staging shared by the C-ABI TUs.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <exception>
#include <cstdlib>
#include <mutex>
#include <new>
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
    unsigned lds_per_wg = 0;          // hipDeviceAttributeMaxSharedMemoryPerBlock of THIS context's device (queried on first use)
    // kernel timer: event pairs bound to the dispatch of the NEXT K1 / K3 launch (hipExtLaunchKernelGGL)
    static constexpr int kTimerSlots = 64;
    hipEvent_t tev[2 * kTimerSlots] = {};
    int timer_slot = -1;
    arkmpc_ctx_stats stats = {};      // arkmpc_ctx_get_stats: written under `mu` only
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
// Every entry point that takes a context holds one of these for its whole body (ENTER / ENTER_EC / ENTER_ED).  Besides the lock and the
// device it is the library's "never unwinds across the ABI" guarantee (the callers are Rust gate closures, fabric.rs:841-854: unwinding
// into them is undefined behaviour): errors are status codes, and a C++ exception the body did not foresee -- in practice std::bad_alloc
// from a container -- reaches this destructor while the stack unwinds and ends the process there, as Rust's own allocation failures do.
struct CtxGuard {
    arkmpc_ctx* c;
    std::unique_lock<std::mutex> lk;
    int rc = ARKMPC_OK;
    int exc_in_flight;
    explicit CtxGuard(arkmpc_ctx* ctx) : c(ctx), lk(ctx->mu), exc_in_flight(std::uncaught_exceptions()) {
        hipError_t e = hipSetDevice(ctx->device);
        if (e != hipSuccess) { ark_set_err(ctx, std::string("hipSetDevice: ") + hipGetErrorString(e)); rc = ARKMPC_ERR_HIP; }
    }
    ~CtxGuard() {
        if (std::uncaught_exceptions() > exc_in_flight) {
            fputs("arkmpc: C++ exception inside the library (out of host memory?); aborting instead of unwinding into the caller\n", stderr);
            std::abort();
        }
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

// ---------------------------------------------------------------------------------------------------------------------------------
// Host link: what the streaming host-buffer paths share (Stage below, the sessions of arkmpc_stream.inc).
// Over PCIe a host-buffer call is bound by the link (~56 GB/s per direction on this box, probes/pcie_probe.hip), so the job is to keep
// the upload direction busy and hide the kernels and the downloads under it: three streams per context -- `up` (H2D DMA), the compute
// stream, `down` (D2H DMA) -- ordered by events only, and the caller's buffers pinned in place so that the copies are true DMA.
// ---------------------------------------------------------------------------------------------------------------------------------
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <iterator>
#include <map>
static inline int link_ensure(arkmpc_ctx* ctx) {
    if (ctx->up) return ARKMPC_OK;
    ARK_HIP(ctx, hipStreamCreateWithFlags(&ctx->up, hipStreamNonBlocking));
    ARK_HIP(ctx, hipStreamCreateWithFlags(&ctx->down, hipStreamNonBlocking));
    return ARKMPC_OK;
}

// Process-wide registry of the host ranges pinned THROUGH THIS LIBRARY (hipHostRegister): by a session / import for the duration of a call
// (library references) and by the caller's arkmpc_host_register (caller references).  Reference-counted: two sessions may name the same buffer
// (a party's out_de is its in-process peer's peer_de; batch_mul(&a, &a) passes one vector twice), and the pinning must outlive the last DMA
// of either.  ROCm accepts a second hipHostRegister of a registered range and the first hipHostUnregister then drops it for both, so ranges
// pinned by somebody else (arkmpc_host_alloc, torch's pinned tensors) are detected up front (hipPointerGetAttributes) and left alone.
// true if the HIP runtime already tracks the address (pinned / registered host memory, device memory); plain malloc memory is reported
// either as an error (older runtimes) or as hipMemoryTypeUnregistered
static inline bool runtime_knows(const void* p) {
    hipPointerAttribute_t attr;
    const hipError_t e = hipPointerGetAttributes(&attr, p);
    if (e != hipSuccess) { (void)hipGetLastError(); return false; }
    return attr.type != hipMemoryTypeUnregistered;
}
// a whole range: its first and its last byte (a range that only begins inside somebody's registration is not pinned memory)
static inline bool runtime_knows_range(const void* p, size_t bytes) {
    return bytes && runtime_knows(p) && runtime_knows((const char*)p + bytes - 1);
}
// Does [p, p + bytes) lie inside ONE allocation / registration the runtime tracks?  (Round-5 advisor finding: a vector that starts in one
// registration and ends in another, with a gap between them, passed the end-point check, and a kernel reading it through the first range's
// device alias would fault -- XNACK is off on this platform.)  A runtime that cannot name the range falls back to the end-point check.
static inline bool one_runtime_range(const void* p, size_t bytes) {
    void* start = nullptr;
    size_t size = 0;
    if (hipPointerGetAttribute(&start, HIP_POINTER_ATTRIBUTE_RANGE_START_ADDR, (hipDeviceptr_t)p) == hipSuccess &&
        hipPointerGetAttribute(&size, HIP_POINTER_ATTRIBUTE_RANGE_SIZE, (hipDeviceptr_t)p) == hipSuccess && start && size)
        return (uintptr_t)p >= (uintptr_t)start && (uintptr_t)p + bytes <= (uintptr_t)start + size;
    (void)hipGetLastError();
    return runtime_knows((const char*)p + bytes - 1);
}
// Devices this process has (had) a context on: what drain_after_registration() below waits for.
inline std::atomic<unsigned>& devices_in_use() { static std::atomic<unsigned> m{0}; return m; }
#ifdef ARKMPC_HAZARD_SWITCHES
// WAIT AFTER REGISTRATION (round 5; only in the hazard build, tools/crash_hunt.sh: ARKMPC_PIN_DRAIN=1, =2 adds a map + unmap of a page of our
// own).  Chasing a flaky soak showed that a KERNEL which addresses in place a caller's vector that this library registered itself can read stale
// memory in a few of the vector's pages when the vector's address had an earlier registered life (freed, handed out again by malloc with other
// physical pages) -- profiles/r05_hazard/README.md.  Waiting for the device after every registration shrank that window without closing it; what
// closed it is not letting kernels address such vectors at all (Place::zc below).  The wait is kept as a switch for the probe.
static inline void drain_after_registration() {
    static const int mode = getenv("ARKMPC_PIN_DRAIN") ? atoi(getenv("ARKMPC_PIN_DRAIN")) : 0;
    if (!mode) return;
    int cur = 0;
    const bool have_cur = hipGetDevice(&cur) == hipSuccess;
    const unsigned mask = devices_in_use().load(std::memory_order_relaxed);
    bool moved = false;
    for (int d = 0; d < 32; ++d) {
        if (!(mask >> d & 1u)) continue;
        if (have_cur && d != cur) { if (hipSetDevice(d) != hipSuccess) continue; moved = true; }
        (void)hipDeviceSynchronize();
    }
    if (!mask) (void)hipDeviceSynchronize();
    if (moved && have_cur) (void)hipSetDevice(cur);
    if (mode == 2) {
        static void* page = aligned_alloc(4096, 4096);
        if (page && hipHostRegister(page, 4096, hipHostRegisterDefault) == hipSuccess) (void)hipHostUnregister(page);
    }
    (void)hipGetLastError();
}
#else
static inline void drain_after_registration() {}
#endif
struct PinRegistry {
    std::mutex mu;
    std::condition_variable cv;
    // busy: a hipHostRegister / hipHostUnregister of this range is in flight OUTSIDE the lock (0.2-0.7 ms per 64 MiB: two in-process parties,
    // or group members in threads, pin their vectors side by side; a second asker of the SAME range waits on cv for the outcome).
    // caller_refs: references held through arkmpc_host_register -- while there is one the range is the caller's pinned memory, not a per-call
    // registration of the library.  reused: the range overlaps addresses that already had a registered life which ended (see `retired`).
    struct Ent { size_t bytes; int refs; int caller_refs; bool reused; bool busy; };
    std::map<uintptr_t, Ent> ents;
    using It = std::map<uintptr_t, Ent>::iterator;
    // Addresses whose registration (through this library) has ENDED, as merged page-granular intervals [lo, hi).  A later registration that
    // overlaps one is marked `reused`, and no kernel addresses such a vector in place (Place::zc): the stale-read hazard of DESIGN section 4
    // needs a kernel reading through a mapping made at an address that had an earlier registered life.  Whatever a caller does -- register and
    // unregister per gate on fresh Vecs that malloc hands out again -- zero-copy kernels only ever see addresses in their FIRST registered life.
    std::map<uintptr_t, uintptr_t> retired;
    static constexpr size_t kMaxRetired = 4096;
    static constexpr uintptr_t kPage = 4096;

    It containing(uintptr_t a, size_t bytes) {             // (mu held)
        It it = ents.upper_bound(a);
        if (it == ents.begin()) return ents.end();
        --it;
        return (a >= it->first && a + bytes <= it->first + it->second.bytes) ? it : ents.end();
    }
    bool overlaps_entry(uintptr_t a, size_t bytes) {       // (mu held) any entry that shares a byte with [a, a + bytes)
        It it = ents.lower_bound(a + bytes);
        if (it == ents.begin()) return false;
        --it;
        return it->first + it->second.bytes > a;
    }
    It settled(std::unique_lock<std::mutex>& lk, uintptr_t a, size_t bytes) {    // the containing entry once no runtime call is in flight for it
        for (;;) {
            It it = containing(a, bytes);
            if (it == ents.end() || !it->second.busy) return it;
            cv.wait(lk);
        }
    }
    void retire(uintptr_t lo, size_t bytes) {              // (mu held)
        uintptr_t hi = (lo + bytes + kPage - 1) & ~(kPage - 1);
        lo &= ~(kPage - 1);
        auto it = retired.upper_bound(lo);
        if (it != retired.begin()) {
            auto pr = std::prev(it);
            if (pr->second >= lo) { lo = pr->first; hi = std::max(hi, pr->second); it = retired.erase(pr); }
        }
        while (it != retired.end() && it->first <= hi) { hi = std::max(hi, it->second); it = retired.erase(it); }
        retired[lo] = hi;
        while (retired.size() > kMaxRetired) {             // bounded: close the smallest gap (the set only ever grows, which errs on the safe side)
            auto best = retired.begin();
            uintptr_t gap = ~(uintptr_t)0;
            for (auto a = retired.begin(), b = std::next(a); b != retired.end(); ++a, ++b)
                if (b->first - a->second < gap) { gap = b->first - a->second; best = a; }
            auto nx = std::next(best);
            best->second = nx->second;
            retired.erase(nx);
        }
    }
    bool retired_overlaps(uintptr_t a, size_t bytes) {     // (mu held)
        auto it = retired.upper_bound(a);
        if (it != retired.begin() && std::prev(it)->second > a) return true;
        return it != retired.end() && it->first < a + bytes;
    }
    // A reference for [p, p + bytes): on the entry that contains it, or on a new registration.  Returns the entry's base, or 0 if the range is
    // not ours to pin (pinned by somebody else, not host memory, straddles an entry) or could not be pinned.  *performed = a hipHostRegister ran.
    uintptr_t acquire(const void* p, size_t bytes, bool caller = false, bool* performed = nullptr) {
        const uintptr_t a = (uintptr_t)p;
        if (performed) *performed = false;
        std::unique_lock<std::mutex> lk(mu);
        It it = settled(lk, a, bytes);
        if (it != ents.end()) { it->second.refs++; if (caller) it->second.caller_refs++; return it->first; }
        if (overlaps_entry(a, bytes)) return 0;                                    // begins or ends inside one of our registrations: leave it to the pageable path
        ents[a] = Ent{bytes, 1, caller ? 1 : 0, false, true};                      // reserved; the runtime is asked without the lock
        lk.unlock();
        bool ok = !runtime_knows_range(p, bytes);                                  // pinned by the caller already (or not host memory)
        if (ok && hipHostRegister(const_cast<void*>(p), bytes, hipHostRegisterDefault) != hipSuccess) { (void)hipGetLastError(); ok = false; }
        if (ok) drain_after_registration();
        lk.lock();
        it = ents.find(a);
        if (!ok) { ents.erase(it); cv.notify_all(); return 0; }
        it->second.busy = false;
        it->second.reused = retired_overlaps(a, bytes);
        cv.notify_all();
        if (performed) *performed = true;
        return a;
    }
    // the same for a range already known to be pinned (by the caller, or by an entry of this registry): a reference if it is ours, no runtime calls
    uintptr_t acquire_if_ours(const void* p, size_t bytes) {
        std::unique_lock<std::mutex> lk(mu);
        It it = settled(lk, (uintptr_t)p, bytes);
        if (it == ents.end()) return 0;
        it->second.refs++;
        return it->first;
    }
    // is [p, p + bytes) inside ONE range registered through this library (no reference taken)?  *ours: held by the library's own per-call
    // registrations only (no arkmpc_host_register on it); *reused: its addresses had an earlier registered life.
    bool lookup(const void* p, size_t bytes, bool* ours, bool* reused) {
        std::lock_guard<std::mutex> lk(mu);
        It it = containing((uintptr_t)p, bytes);
        if (it == ents.end()) return false;
        *ours = it->second.caller_refs == 0;
        *reused = it->second.reused;
        return true;
    }
    void release(uintptr_t base, bool caller = false) {
        std::unique_lock<std::mutex> lk(mu);
        It it = ents.find(base);
        if (it == ents.end()) return;
        if (caller && it->second.caller_refs > 0) it->second.caller_refs--;
        if (--it->second.refs > 0) return;
        it->second.busy = true;                            // nobody takes a reference while the runtime drops the mapping
        const size_t bytes = it->second.bytes;
        lk.unlock();
        (void)hipHostUnregister((void*)base);
        (void)hipGetLastError();
        lk.lock();
        ents.erase(base);
        retire(base, bytes);
        cv.notify_all();
    }
    size_t retired_intervals() { std::lock_guard<std::mutex> lk(mu); return retired.size(); }
};
// one registry per process (the engine is one shared object; inline + function-local static = one instance across its translation units)
inline PinRegistry& pin_registry() { static PinRegistry r; return r; }

// pins caller buffers in place for the lifetime of the object.  A buffer that cannot be pinned (read-only mapping, too small to matter: the
// copy then takes the runtime's pageable path, slower but correct) is not an error.
struct HostPins {
    std::vector<uintptr_t> held;
    static size_t min_bytes() { static const size_t v = getenv("ARKMPC_PIN_MIN_KB") ? (size_t)atoll(getenv("ARKMPC_PIN_MIN_KB")) << 10 : (size_t)1 << 20; return v; }
    // Per-call registration of a caller's pageable vector, for the DMA copy pipeline (true asynchronous DMAs instead of the runtime's blocking
    // pageable copies: 8.2 instead of 13.2 ms per 2^20-gate session).  On by default; what is NOT done any more with such a vector is letting a
    // kernel address it in place (Place::zc).  ARKMPC_PIN_IN_PLACE=0 or ARKMPC_NO_PIN=1: never register (the runtime's pageable copies).
    static bool in_place() {
        static const bool off = (getenv("ARKMPC_PIN_IN_PLACE") && getenv("ARKMPC_PIN_IN_PLACE")[0] == '0') || (getenv("ARKMPC_NO_PIN") && getenv("ARKMPC_NO_PIN")[0] == '1');
        return !off;
    }
    void pin(const void* p, size_t bytes) {
        if (!in_place() || !p || bytes < min_bytes()) return;
        const uintptr_t base = pin_registry().acquire(p, bytes);
        if (base) held.push_back(base);
    }
    void keep(const void* p, size_t bytes) {               // p is pinned already: only make sure it stays so if the pin is one of ours
        const uintptr_t base = pin_registry().acquire_if_ours(p, bytes);
        if (base) held.push_back(base);
    }
    void release() {
        for (uintptr_t b : held) pin_registry().release(b);
        held.clear();
    }
    ~HostPins() { release(); }
};

// Where a caller's vector lives (streaming sessions, group transfers, asynchronous batch imports).  Every vector is looked at on its own: a circuit keeps x, y (the previous gates' outputs) and its result in
// HBM while the triples come from a preprocessing source in host memory and the payloads cross a host link (fabric.rs:894-915).
enum class Mem { Pageable, Pinned, Device, Foreign };
struct Place {
    Mem kind = Mem::Pageable;
    void* dev = nullptr;                                   // what a kernel dereferences: the pointer itself (Device), its mapped alias (Pinned)
    bool ours = false;                                     // pinned because THIS LIBRARY registered it in place (per call), not because the caller holds it in pinned memory
    bool reused = false;                                   // registered (through this library) at addresses that had an earlier registered life
    // may a kernel address the vector where it lies?  Not a vector this library registered itself, and not one registered at a retired address:
    // kernels that read or write such a vector in place are what the stale-memory hazard of DESIGN section 4 needs (30 stress repetitions with
    // them: up to 27 wrong vectors; the same registrations used by DMA only: none) -- those vectors travel by DMA.
#ifdef ARKMPC_HAZARD_SWITCHES
    static bool zc_on_own_pins() { static const bool on = getenv("ARKMPC_ZC_ON_OWN_PINS") && getenv("ARKMPC_ZC_ON_OWN_PINS")[0] == '1'; return on; }   // (hazard build only: tools/crash_hunt.sh)
#else
    static constexpr bool zc_on_own_pins() { return false; }
#endif
    bool zc() const { return dev && !((uintptr_t)dev & 15) && (kind == Mem::Device || (kind == Mem::Pinned && ((!ours && !reused) || zc_on_own_pins()))); }    // (the zero-copy kernels move 16-byte quarters; a Rust Vec only promises 8)
    bool device() const { return kind == Mem::Device; }
};
static inline Place classify(arkmpc_ctx* ctx, const void* p, size_t bytes) {
    Place pl;
    if (!p || !bytes) return pl;
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, p) != hipSuccess) { (void)hipGetLastError(); return pl; }       // plain malloc memory on older runtimes
    if (attr.type == hipMemoryTypeDevice) {
        pl.kind = attr.device == ctx->device ? Mem::Device : Mem::Foreign;
        pl.dev = const_cast<void*>(p);
        return pl;
    }
    if (attr.type != hipMemoryTypeHost || !attr.devicePointer) return pl;                               // hipMemoryTypeUnregistered, managed
    bool ours = false, reused = false;
    if (!pin_registry().lookup(p, bytes, &ours, &reused) && !one_runtime_range(p, bytes)) return pl;    // begins in one registration and ends outside it: not pinned memory
    pl.kind = Mem::Pinned;
    pl.dev = attr.devicePointer;
    pl.ours = ours;
    pl.reused = reused;
    // a caller-held vector that would otherwise be addressed in place (atomic: group members classify through member 0's context from their own threads)
    if (reused && !ours && !((uintptr_t)pl.dev & 15)) __atomic_fetch_add(&ctx->stats.zc_refused_reused_address, 1, __ATOMIC_RELAXED);
    return pl;
}
// The same for a vector a kernel is going to address in place: the registry reference is taken FIRST (HostPins::keep), the pointer is looked
// at afterwards.  The other order left a window -- round-4 advisor finding: a vector that is pinned only because ANOTHER session of this library
// registered it (an in-process peer's out_de passed as this party's peer_de) could be unregistered by that session between the look and the
// reference, and the kernel would then fault on an unmapped address.  If the look says "not pinned after all" the reference is dropped again.
static inline Place classify_and_hold(arkmpc_ctx* ctx, HostPins& pins, const void* p, size_t bytes) {
    const size_t before = pins.held.size();
    pins.keep(p, bytes);
    const Place pl = classify(ctx, p, bytes);
    if (pl.kind != Mem::Pinned && pins.held.size() > before) { pin_registry().release(pins.held.back()); pins.held.pop_back(); }
    return pl;
}
// the events of one streamed call: taken from the context's free list, returned when the call has drained
struct LinkEvents {
    arkmpc_ctx* ctx;
    std::vector<hipEvent_t> used;
    explicit LinkEvents(arkmpc_ctx* c) : ctx(c) {}
    hipEvent_t take() {
        hipEvent_t ev = nullptr;
        if (!ctx->link_ev.empty()) { ev = ctx->link_ev.back(); ctx->link_ev.pop_back(); }
        else if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        used.push_back(ev);
        return ev;
    }
    // `to` waits (on the device) for everything submitted to `from` so far.  (Either stream may be the NULL stream -- a context bound to
    // torch's default stream has ctx->stream == nullptr -- so "no waiter" is its own entry point, mark(), not a null `to`.)
    int edge(hipStream_t from, hipStream_t to) {
        hipEvent_t ev = take();
        if (!ev) { ark_set_err(ctx, "hipEventCreate failed"); return ARKMPC_ERR_HIP; }
        ARK_HIP(ctx, hipEventRecord(ev, from));
        ARK_HIP(ctx, hipStreamWaitEvent(to, ev, 0));
        return ARKMPC_OK;
    }
    // an event the HOST can query / wait for: everything submitted to `from` so far
    int mark(hipStream_t from, hipEvent_t* out_ev) {
        hipEvent_t ev = take();
        if (!ev) { ark_set_err(ctx, "hipEventCreate failed"); return ARKMPC_ERR_HIP; }
        ARK_HIP(ctx, hipEventRecord(ev, from));
        *out_ev = ev;
        return ARKMPC_OK;
    }
    void give_back() {                     // only after the streams that recorded / waited on them have drained
        for (hipEvent_t e : used) ctx->link_ev.push_back(e);
        used.clear();
    }
};

// n arkworks ScalarShare records at rec_dev (a device-side address: HBM, or the mapped alias of pinned host memory) -> split columns, one pass,
// enqueued on `st` (csrc/arkmpc_batch.hip)
int ark_import_split(hipStream_t st, size_t n, const void* rec_dev, uint64_t* share_col, uint64_t* mac_col);
// the wire codec on device buffers, for callers that hold the context lock (csrc/arkmpc_wire.hip)
int ark_wire_encode_scalars_device(arkmpc_ctx* ctx, uint64_t result_id, size_t n, const uint64_t* d_scalars, uint8_t* d_frame, size_t cap, size_t* out_len);
int ark_wire_decode_scalars_device(arkmpc_ctx* ctx, const uint8_t* d_frame, size_t frame_len, size_t max_n, uint64_t* d_scalars, size_t* out_n, uint64_t* out_result_id);

// Chunk schedule of the chunked array of a phase: [lo, lo + cnt) ranges covering n.  Full chunks are 2^18 elements (16 MiB of 64-byte
// records: the DMA engines reach 54 of their 56 GB/s at that size, 49 at 4 MiB); the tail tapers by halves down to 2^16 so that what is left
// AFTER the last upload byte (one kernel + one download of the last chunk) is short; small batches run in about four chunks, never below 2^14.
static inline std::vector<std::pair<size_t, size_t>> stream_chunks(size_t n) {
    static const int env = getenv("ARKMPC_STREAM_CHUNK_LOG2") ? atoi(getenv("ARKMPC_STREAM_CHUNK_LOG2")) : 0;
    static const bool taper = !(getenv("ARKMPC_STREAM_TAPER") && getenv("ARKMPC_STREAM_TAPER")[0] == '0');
    size_t full = (size_t)1 << 18;
    if (env >= 10 && env <= 28) full = (size_t)1 << env;
    else while (full > ((size_t)1 << 14) && full * 4 > n) full >>= 1;
    const size_t cmin = full >= ((size_t)1 << 18) ? full >> 2 : full;
    std::vector<std::pair<size_t, size_t>> v;
    for (size_t lo = 0; lo < n;) {
        const size_t rem = n - lo;
        size_t cnt = full;
        if (taper && rem < 2 * full) { cnt = (rem / 2 + 255) & ~(size_t)255; if (cnt < cmin) cnt = cmin; }
        if (cnt > rem || rem - cnt < cmin / 4) cnt = rem;
        v.emplace_back(lo, cnt);
        lo += cnt;
    }
    return v;
}

// Staging of one API call.  In device mode `in`/`out` are pass-through (with an alignment check); in host mode they carve device
// buffers out of the arena, upload inputs and remember outputs.
//
// Two protocols.  (1) Whole-batch: declare_*, commit(), launch on in<>()/out<>(), finish() -- all uploads, the kernels, all downloads,
// one after the other on the context's stream.  (2) ELEMENTWISE ops (element i of every output depends on element i of every input)
// add elementwise(n) before commit() and launch per chunk:
//        size_t lo, cnt;  while (st.next_chunk(&lo, &cnt)) { launch on elements [lo, lo + cnt) }  return st.finish();
// In device mode, for small host batches and for PAGEABLE host buffers that is one chunk (0, n): the runtime's own pageable copy (it pins
// and DMAs piecewise, 54-56 GB/s on this box) is as fast as anything a single call can do -- pinning the buffers in place first costs
// about 60 % of the DMA it would speed up, which eats the one gain on offer, the hidden downloads (measured: 13.8 ms pinned per call
// vs 13.3 ms serial for K1 + K2+K3 at 2^20 gates).  For large batches in buffers the caller has ALREADY pinned (arkmpc_host_register /
// arkmpc_host_alloc) the copies are true asynchronous DMA: commit() sends every input but the last up whole on the `up` stream;
// next_chunk() uploads the last input chunk by chunk, the caller's launch for chunk k runs behind it on the compute stream and the
// outputs of chunk k go down on the `down` stream while chunk k+1 is still going up.  Buffers of `segs` segments of n elements
// (d||e: 2) are chunked per segment.
struct Stage {
    arkmpc_ctx* ctx;
    int rc = ARKMPC_OK;
    struct In { const void* host; size_t bytes; size_t off; unsigned segs; };
    std::vector<In> ins;
    struct OutPlan { void* host; size_t bytes; size_t off; unsigned segs; };
    std::vector<OutPlan> oplan;
    size_t total = 0;
    explicit Stage(arkmpc_ctx* c) : ctx(c), evs(c) {}
    Stage(const Stage&) = delete;
    Stage& operator=(const Stage&) = delete;
    ~Stage() { if (streamed && !drained) drain(); }

    static size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }

    // pass 1 (host mode): declare buffers; pass 2: resolve.  To keep call sites linear we do a
    // two-phase protocol: declare_* returns an index, then commit() uploads, and ptr(i) resolves.
    int declare_in(const void* p, size_t bytes, unsigned segs = 1) {
        if (!p && bytes) { rc = ark_bad(ctx, "null input pointer"); }
        ins.push_back({p, bytes, total, segs}); total += align_up(bytes);
        return (int)ins.size() - 1;
    }
    int declare_out(void* p, size_t bytes, unsigned segs = 1) {
        if (!p && bytes) { rc = ark_bad(ctx, "null output pointer"); }
        oplan.push_back({p, bytes, total, segs}); total += align_up(bytes);
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

    // ---- elementwise protocol ---------------------------------------------------------------------------------------------------
    bool ew = false, streamed = false, drained = false, chunk_open = false;
    size_t ew_n = 0, chunk_i = 0, cur_lo = 0, cur_cnt = 0;
    std::vector<std::pair<size_t, size_t>> chunks;
    LinkEvents evs;
    void elementwise(size_t n) { ew = true; ew_n = n; }
    static size_t stream_min_bytes() {       // below this the whole-batch protocol is as fast and has less to set up
        static const size_t v = getenv("ARKMPC_STREAM_MIN_MB") ? (size_t)atoll(getenv("ARKMPC_STREAM_MIN_MB")) << 20 : (size_t)8 << 20;
        return v;
    }
    static bool stream_enabled() { static const bool on = !(getenv("ARKMPC_NO_STREAM") && getenv("ARKMPC_NO_STREAM")[0] == '1'); return on; }

    int fail_hip(hipError_t e, const char* what) { ark_set_err(ctx, std::string(what) + ": " + hipGetErrorString(e)); return rc = ARKMPC_ERR_HIP; }
    // one chunk [lo, lo + cnt) of a declared buffer, segment by segment
    template <class B> int copy_chunk(const B& b, size_t lo, size_t cnt, bool up) {
        const size_t eb = b.bytes / ((size_t)b.segs * ew_n);
        for (unsigned sg = 0; sg < b.segs; ++sg) {
            const size_t o = ((size_t)sg * ew_n + lo) * eb;
            hipError_t e = up ? hipMemcpyAsync(ctx->arena + b.off + o, (const char*)b.host + o, cnt * eb, hipMemcpyHostToDevice, ctx->up)
                              : hipMemcpyAsync((char*)b.host + o, ctx->arena + b.off + o, cnt * eb, hipMemcpyDeviceToHost, ctx->down);
            if (e != hipSuccess) return fail_hip(e, up ? "H2D" : "D2H");
        }
        return ARKMPC_OK;
    }
    int commit_streamed() {
        if ((rc = arena_reserve(ctx, total + scratch_total))) return rc;
        if ((rc = link_ensure(ctx))) return rc;
        streamed = true;
        if ((rc = evs.edge(ctx->stream, ctx->up))) return rc;        // the arena's last user ran on the compute stream
        for (size_t i = 0; i + 1 < ins.size(); ++i) {
            hipError_t e = hipMemcpyAsync(ctx->arena + ins[i].off, ins[i].host, ins[i].bytes, hipMemcpyHostToDevice, ctx->up);
            if (e != hipSuccess) return fail_hip(e, "H2D");
        }
        chunks = stream_chunks(ew_n);
        return ARKMPC_OK;
    }
    int close_chunk() {
        chunk_open = false;
        hipError_t le = hipGetLastError();
        if (le != hipSuccess) return fail_hip(le, "kernel launch");
        if ((rc = evs.edge(ctx->stream, ctx->down))) return rc;
        for (auto& o : oplan) if (copy_chunk(o, cur_lo, cur_cnt, false)) return rc;
        return ARKMPC_OK;
    }
    bool next_chunk(size_t* lo, size_t* cnt) {
        if (rc) return false;
        if (!streamed) {                                    // device pointers, or a host batch staged whole: one chunk
            if (chunk_i++ == 0 && ew_n) { *lo = 0; *cnt = ew_n; return true; }
            return false;
        }
        if (chunk_open && close_chunk()) return false;
        if (chunk_i == chunks.size()) return false;
        cur_lo = chunks[chunk_i].first; cur_cnt = chunks[chunk_i].second; ++chunk_i;
        if (!ins.empty() && copy_chunk(ins.back(), cur_lo, cur_cnt, true)) return false;
        if ((rc = evs.edge(ctx->up, ctx->stream))) return false;
        chunk_open = true;
        *lo = cur_lo; *cnt = cur_cnt;
        return true;
    }
    int drain() {
        drained = true;
        hipError_t e1 = hipStreamSynchronize(ctx->up), e2 = hipStreamSynchronize(ctx->stream), e3 = hipStreamSynchronize(ctx->down);
        evs.give_back();
        if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) {
            if (!rc) fail_hip(e1 != hipSuccess ? e1 : (e2 != hipSuccess ? e2 : e3), "streamed call");
        }
        return rc;
    }

    int commit() {
        if (rc) return rc;
        if (ew && ctx->host_buffers && stream_enabled() && ew_n && total >= stream_min_bytes()) {
            bool go = true;                                  // every buffer: segs x n elements of a whole number of bytes, and pinned by the caller
            for (auto& i : ins) go = go && i.bytes % ((size_t)i.segs * ew_n) == 0 && (i.bytes < ((size_t)1 << 20) || runtime_knows_range(i.host, i.bytes));
            for (auto& o : oplan) go = go && o.bytes % ((size_t)o.segs * ew_n) == 0 && (o.bytes < ((size_t)1 << 20) || runtime_knows_range(o.host, o.bytes));
            if (go) return commit_streamed();
        }
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
        if (streamed) {
            if (!rc && chunk_open) close_chunk();
            return drain();
        }
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
    // of the streaming kernels: tools/k3_occupancy_probe.sh @77e688c); 0 / unset in production
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
