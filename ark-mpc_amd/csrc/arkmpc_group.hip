// arkmpc_group.hip -- the multi-GPU form of the path INSIDE the C ABI: ONE process drives N devices of one node.
//
// Why it exists: a party of the reference is ONE process (`MpcFabric::new` spawns its executor and network threads inside the caller's
// process, fabric.rs:402-466); it cannot be re-launched once per GPU.  A patched `batch_mul` / `open_authenticated_batch` closure that
// wants all 8 GPUs of an MI355X node therefore needs the sharding behind the FFI, not in a launcher.
//
// Model: a batch of n independent gates is cut into contiguous index ranges [g*n/G, (g+1)*n/G) -- the same
// ranges as ark-mpc_amd/sharding.py -- one per member; every member owns a context (its device, its stream) and runs the ordinary
// single-device kernels on its range.  There is NO collective in the arithmetic.  What crosses devices:
//   * gathers of opened-value / MAC-check / d||e buffers, as DIRECT PEER WRITES over xGMI: every source device pushes its range into
//     the destination buffer (hipMemcpyPeerAsync on the source's stream, or -- the *_gathered forms -- the producing kernel itself
//     storing through the peer mapping), so an all-gather keeps all 7 links of every GPU busy at once instead of walking a ring;
//   * the one-word MAC-verify flag, AND-reduced on the host from the members' mapped flags;
//   * nothing for the SHA3 commitment: the sponge is sequential (commitment.rs:36-40), so each device converts its own range to
//     big-endian bytes and DMAs it to pinned host memory over its own PCIe link while the host absorbs the ranges in order.
// A "sharded vector" is an array of G device pointers; member g's pointer addresses `segs` consecutive segments of cnt_g elements
// (d||e: 2 segments of Scalars; split-layout ScalarShares: share column then MAC column; everything else: 1 segment).
//
// Device ids may repeat ({0,0,0,0}): members then share a GPU (each with its own stream), which is how the whole path -- peer copies
// included -- is tested bit-for-bit on a single-GPU box.
#include "arkmpc_internal.hpp"
#include <cstdlib>
#include <thread>

struct arkmpc_group {
    int field_id = 0;
    int G = 0;
    std::vector<arkmpc_ctx*> ctx;
    std::vector<int> dev;
    std::vector<unsigned char> peer;          // peer[from * G + to] = 1: `from` may address `to`'s memory
    std::vector<hipEvent_t> ev;               // one per member: "my pushes are done"
    std::vector<hipEvent_t> wev;              // one per member: "what I had queued on the buffer you are about to overwrite is done"
    std::mutex mu;                            // group calls are serialised
    std::mutex err_mu;
    std::string err;
    // commitment pipeline: per member 2 pinned slots + 2 device slots + 2 events
    std::vector<unsigned char*> pin;          // 2 * G
    std::vector<unsigned char*> dstage;       // 2 * G
    std::vector<hipEvent_t> cev;              // 2 * G
};

namespace {
constexpr size_t kCommitChunk = (size_t)1 << 18;     // elements per pipeline slot (8 MiB of bytes)

void gset_err(arkmpc_group* g, const std::string& s) { std::lock_guard<std::mutex> lk(g->err_mu); g->err = s; }
int gbad(arkmpc_group* g, const char* what) { gset_err(g, what); return ARKMPC_ERR_BAD_ARG; }
// status of a member call -> group status (copying the member's message)
int gfail(arkmpc_group* g, int member, int rc, const char* what) {
    gset_err(g, std::string(what) + " (member " + std::to_string(member) + "): " + arkmpc_last_error(g->ctx[member]));
    return rc;
}
#define GHIP(g, call)                                                                                    \
    do {                                                                                                 \
        hipError_t e__ = (call);                                                                         \
        if (e__ != hipSuccess) { gset_err((g), std::string(#call) + ": " + hipGetErrorString(e__)); return ARKMPC_ERR_HIP; } \
    } while (0)
#define GCALL(g, m, call)                                                                                \
    do { int rc__ = (call); if (rc__ != ARKMPC_OK) return gfail((g), (m), rc__, #call); } while (0)

inline void range(const arkmpc_group* g, size_t n, int m, size_t* lo, size_t* cnt) {
    // (n * m) / G without overflow for n < 2^58 (G <= 64)
    const size_t a = (size_t)(((unsigned __int128)n * (unsigned)m) / (unsigned)g->G);
    const size_t b = (size_t)(((unsigned __int128)n * (unsigned)(m + 1)) / (unsigned)g->G);
    *lo = a; *cnt = b - a;
}

// UNREGISTERED host vectors and N links.  A vector the library does not register (below ARKMPC_PIN_MIN_KB, or ARKMPC_PIN_IN_PLACE=0) travels
// as the runtime's pageable copies -- and those run ON THE CALLING THREAD (the runtime stages or pins the memory itself and returns when the copy
// is done).  Issued member after member from one thread they use one link at a time, which is what the round-4 review objected to.  So when the
// members sit on DISTINCT devices, the member calls that touch a pageable vector are made from one short-lived host thread per member: each
// thread blocks in its own member's copies, all links run.  Members that share a device (the one-GPU test box) share a link and gain nothing:
// they keep the single thread.  ARKMPC_GROUP_THREADS=1 forces the threads (the tests do, in a child process), =0 forbids them.  Pinned vectors
// never need this: their member calls only enqueue.
inline bool members_in_threads(const arkmpc_group* g) {
    static const int env = getenv("ARKMPC_GROUP_THREADS") ? atoi(getenv("ARKMPC_GROUP_THREADS")) : -1;
    if (env == 0 || g->G < 2) return false;
    if (env > 0) return true;
    for (int m = 1; m < g->G; ++m) if (g->dev[m] != g->dev[0]) return true;
    return false;
}
inline bool pageable_host(const arkmpc_group* g, const void* p, size_t bytes) {
    return p && bytes && classify(g->ctx[0], p, bytes).kind == Mem::Pageable;
}
// fn(m) for every member: in G threads (joined before returning) or in a loop.  fn reports through its own per-member slots.
template <class F> void for_members(arkmpc_group* g, bool threads, F fn) {
    int started = 0;
    std::vector<std::thread> th;
    if (threads) {
        try {
            th.reserve(g->G);
            for (; started < g->G; ++started) th.emplace_back([&fn, started] { fn(started); });
        } catch (...) {}                                  // (no more threads to be had: the rest runs here)
    }
    for (int m = started; m < g->G; ++m) fn(m);
    for (auto& t : th) t.join();
}
inline bool layout_ok(int layout) { return layout == ARKMPC_LAYOUT_AOS || layout == ARKMPC_LAYOUT_SPLIT; }
// share / MAC column view of member m's shard of a ScalarShare vector
struct ShareView { const u64* s; const u64* m; size_t stride; };
inline ShareView share_view(int layout, const u64* base, size_t cnt) {
    if (layout == ARKMPC_LAYOUT_SPLIT) return {base, base + 4 * cnt, 4};
    return {base, base + 4, 8};
}
// copy `bytes` from member `from`'s memory into member `to`'s memory on `from`'s stream (a push over from's link to `to`)
int push(arkmpc_group* g, int from, int to, void* dst, const void* src, size_t bytes) {
    if (!bytes) return ARKMPC_OK;
    hipStream_t st = g->ctx[from]->stream;
    GHIP(g, hipSetDevice(g->dev[from]));
    // ARKMPC_GROUP_FORCE_PEER=1 (test hook): take the peer-copy API even between members that share a device, so that the call the
    // multi-GPU path makes is at least executed on a one-GPU box
    static const bool force_peer = getenv("ARKMPC_GROUP_FORCE_PEER") && getenv("ARKMPC_GROUP_FORCE_PEER")[0] == '1';
    if (g->dev[from] == g->dev[to] && !force_peer) GHIP(g, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, st));
    else GHIP(g, hipMemcpyPeerAsync(dst, g->dev[to], src, g->dev[from], bytes, st));
    return ARKMPC_OK;
}
// write-after-read ordering of a cross-member write: whatever member `to` has already queued (a kernel of the previous round still reading
// the destination buffer) must finish before any source member starts writing into it: every source stream waits for `to`'s stream.
int sources_wait_for(arkmpc_group* g, int to) {
    GHIP(g, hipSetDevice(g->dev[to]));
    GHIP(g, hipEventRecord(g->wev[to], g->ctx[to]->stream));
    for (int m = 0; m < g->G; ++m) if (m != to) { GHIP(g, hipSetDevice(g->dev[m])); GHIP(g, hipStreamWaitEvent(g->ctx[m]->stream, g->wev[to], 0)); }
    return ARKMPC_OK;
}
int shards_ok(arkmpc_group* g, size_t n, const void* const* shards, const char* what) {
    if (!shards) return gbad(g, what);
    for (int m = 0; m < g->G; ++m) {
        size_t lo, cnt; range(g, n, m, &lo, &cnt);
        if (cnt && !shards[m]) return gbad(g, what);
        if ((uintptr_t)shards[m] & 15) return gbad(g, "shard pointer not 16-byte aligned");
    }
    return ARKMPC_OK;
}
}  // namespace

extern "C" {

int arkmpc_group_create(int field_id, int n_devices, const int* device_ids, arkmpc_group** out) {
    if (!out) return ARKMPC_ERR_BAD_ARG;
    *out = nullptr;
    if (n_devices <= 0 || n_devices > 64 || !device_ids) return ARKMPC_ERR_BAD_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return ARKMPC_ERR_NO_DEVICE;
    for (int i = 0; i < n_devices; ++i) if (device_ids[i] < 0 || device_ids[i] >= ndev) return ARKMPC_ERR_BAD_ARG;
    arkmpc_group* g = new arkmpc_group();
    g->field_id = field_id; g->G = n_devices;
    g->dev.assign(device_ids, device_ids + n_devices);
    g->ctx.assign(n_devices, nullptr);
    g->ev.assign(n_devices, nullptr);
    g->wev.assign(n_devices, nullptr);
    g->pin.assign(2 * n_devices, nullptr); g->dstage.assign(2 * n_devices, nullptr); g->cev.assign(2 * n_devices, nullptr);
    g->peer.assign((size_t)n_devices * n_devices, 0);
    int rc = ARKMPC_OK;
    for (int i = 0; i < n_devices && rc == ARKMPC_OK; ++i) {
        rc = arkmpc_ctx_create(field_id, device_ids[i], &g->ctx[i]);
        if (rc == ARKMPC_OK && (hipSetDevice(device_ids[i]) != hipSuccess || hipEventCreateWithFlags(&g->ev[i], hipEventDisableTiming) != hipSuccess ||
                                hipEventCreateWithFlags(&g->wev[i], hipEventDisableTiming) != hipSuccess)) rc = ARKMPC_ERR_HIP;
    }
    // peer mappings between every pair of DISTINCT devices (xGMI is a full mesh inside a node); a refusal is recorded, not fatal:
    // copies then fall back to what hipMemcpyPeerAsync does without a mapping, and the *_gathered forms to local + gather
    for (int a = 0; a < n_devices && rc == ARKMPC_OK; ++a)
        for (int b = 0; b < n_devices; ++b) {
            if (device_ids[a] == device_ids[b]) { g->peer[(size_t)a * n_devices + b] = 1; continue; }
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, device_ids[a], device_ids[b]) != hipSuccess || !can) { (void)hipGetLastError(); continue; }
            if (hipSetDevice(device_ids[a]) != hipSuccess) { rc = ARKMPC_ERR_HIP; break; }
            hipError_t e = hipDeviceEnablePeerAccess(device_ids[b], 0);
            if (e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled) g->peer[(size_t)a * n_devices + b] = 1;
            (void)hipGetLastError();
        }
    if (rc != ARKMPC_OK) { arkmpc_group_destroy(g); return rc; }
    *out = g;
    return ARKMPC_OK;
}

int arkmpc_group_destroy(arkmpc_group* g) {
    if (!g) return ARKMPC_ERR_BAD_ARG;
    for (int i = 0; i < g->G; ++i) {
        if (!g->ctx[i]) continue;
        (void)hipSetDevice(g->dev[i]);
        (void)hipStreamSynchronize(g->ctx[i]->stream);
        for (int s = 0; s < 2; ++s) {
            if (g->pin[2 * i + s]) (void)hipHostFree(g->pin[2 * i + s]);
            if (g->dstage[2 * i + s]) (void)arkmpc_free(g->ctx[i], g->dstage[2 * i + s]);
            if (g->cev[2 * i + s]) (void)hipEventDestroy(g->cev[2 * i + s]);
        }
        if (g->ev[i]) (void)hipEventDestroy(g->ev[i]);
        if (g->wev[i]) (void)hipEventDestroy(g->wev[i]);
    }
    for (int i = 0; i < g->G; ++i) if (g->ctx[i]) arkmpc_ctx_destroy(g->ctx[i]);
    delete g;
    return ARKMPC_OK;
}

int arkmpc_group_size(const arkmpc_group* g) { return g ? g->G : 0; }
int arkmpc_group_device(const arkmpc_group* g, int member) { return (g && member >= 0 && member < g->G) ? g->dev[member] : -1; }
arkmpc_ctx* arkmpc_group_ctx(arkmpc_group* g, int member) { return (g && member >= 0 && member < g->G) ? g->ctx[member] : nullptr; }
int arkmpc_group_peer_access(const arkmpc_group* g, int from, int to) {
    if (!g || from < 0 || to < 0 || from >= g->G || to >= g->G) return 0;
    return g->peer[(size_t)from * g->G + to];
}
int arkmpc_group_shard_range(const arkmpc_group* g, size_t n, int member, size_t* out_lo, size_t* out_count) {
    if (!g || member < 0 || member >= g->G || !out_lo || !out_count) return ARKMPC_ERR_BAD_ARG;
    range(g, n, member, out_lo, out_count);
    return ARKMPC_OK;
}
const char* arkmpc_group_last_error(arkmpc_group* g) {
    if (!g) return "null group";
    static thread_local std::string snapshot;
    { std::lock_guard<std::mutex> lk(g->err_mu); snapshot = g->err; }
    return snapshot.c_str();
}
int arkmpc_group_sync(arkmpc_group* g) {
    if (!g) return ARKMPC_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lk(g->mu);
    for (int m = 0; m < g->G; ++m) GCALL(g, m, arkmpc_sync(g->ctx[m]));
    return ARKMPC_OK;
}

// Ordering between TWO groups built on the same devices (the two parties of an in-process run hand each other shard pointers member by
// member): member m of `grp` waits -- on the device -- for everything member m of `producer` has submitted so far.  One call per edge of
// the protocol: after the producer's K1 and before this group's K2+K3 (read after write of the d||e shards), and after this group's K2+K3
// before the producer's next K1 overwrites them (write after read).
int arkmpc_group_wait_group(arkmpc_group* g, arkmpc_group* producer) {
    if (!g || !producer) return ARKMPC_ERR_BAD_ARG;
    if (g == producer) return ARKMPC_OK;
    if (g->G != producer->G) return gbad(g, "group_wait_group: the groups differ in size");
    for (int m = 0; m < g->G; ++m) if (g->dev[m] != producer->dev[m]) return gbad(g, "group_wait_group: member devices differ");
    std::lock_guard<std::mutex> lk(g->mu);                 // (the event belongs to the waiter: two waiters on one producer do not share it)
    for (int m = 0; m < g->G; ++m) {
        GHIP(g, hipSetDevice(g->dev[m]));
        GHIP(g, hipEventRecord(g->wev[m], producer->ctx[m]->stream));
        GHIP(g, hipStreamWaitEvent(g->ctx[m]->stream, g->wev[m], 0));
    }
    return ARKMPC_OK;
}

// ---- sharded vectors -------------------------------------------------------------------------------------------------------------
int arkmpc_group_malloc(arkmpc_group* g, size_t n, size_t segs, size_t elem_words, uint64_t** out_shards) {
    if (!g) return ARKMPC_ERR_BAD_ARG;
    if (!out_shards || !segs || !elem_words || segs > 64 || elem_words > 64) return gbad(g, "group_malloc: bad arguments");
    if (n > (((size_t)1 << 50) / (segs * elem_words))) return gbad(g, "group_malloc: batch too large");
    std::lock_guard<std::mutex> lk(g->mu);
    for (int m = 0; m < g->G; ++m) out_shards[m] = nullptr;
    for (int m = 0; m < g->G; ++m) {
        size_t lo, cnt; range(g, n, m, &lo, &cnt);
        void* p = nullptr;
        int rc = arkmpc_malloc(g->ctx[m], (cnt ? cnt : 1) * segs * elem_words * 8, &p);
        if (rc) {
            for (int k = 0; k < m; ++k) { (void)arkmpc_free(g->ctx[k], out_shards[k]); out_shards[k] = nullptr; }
            return gfail(g, m, rc, "arkmpc_malloc");
        }
        out_shards[m] = (uint64_t*)p;
    }
    return ARKMPC_OK;
}
int arkmpc_group_free(arkmpc_group* g, uint64_t* const* shards) {
    if (!g) return ARKMPC_ERR_BAD_ARG;
    if (!shards) return ARKMPC_OK;
    std::lock_guard<std::mutex> lk(g->mu);
    int rc = ARKMPC_OK;
    for (int m = 0; m < g->G; ++m) if (shards[m]) { int r = arkmpc_free(g->ctx[m], shards[m]); if (r && !rc) rc = gfail(g, m, r, "arkmpc_free"); }
    return rc;
}

// host vector of `segs` segments x n elements  <->  shards.  The host vector is pinned ONCE, whole (HostPins: nothing to do for a vector
// the caller registered or allocated pinned), so that every member's copy is a true asynchronous DMA: the G copies are all enqueued -- each
// on its member's stream, i.e. on its own GPU's PCIe link -- before the first one is waited for.  (Until round 5 the copies were issued from
// pageable memory: the runtime stages such a copy through its own bounce buffers on the calling thread, so the members' uploads could
// serialise behind one another.)  Vectors below ARKMPC_PIN_MIN_KB are not worth a registration and travel as pageable copies.  Returns when all
// copies have landed.
static int host_xfer(arkmpc_group* g, bool to_device, size_t n, size_t segs, size_t ew, const u64* host_c, u64* host_m, u64* const* shards) {
    if (!g) return ARKMPC_ERR_BAD_ARG;
    if (!segs || !ew) return gbad(g, "bad segment / element size");
    if (n && !(to_device ? (const void*)host_c : (const void*)host_m)) return gbad(g, "null host pointer");
    int rc = shards_ok(g, n, (const void* const*)shards, "null shard pointer");
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(g->mu);
    HostPins pins;
    const void* hp = to_device ? (const void*)host_c : (const void*)host_m;
    if (n) pins.pin(hp, segs * n * ew * 8);
    if (n && members_in_threads(g) && pageable_host(g, hp, segs * n * ew * 8)) {
        // pageable copies block their caller: one thread per member, so that every member's link is busy (see members_in_threads)
        std::vector<int> bad(g->G, 0);
        for_members(g, true, [&](int m) {
            size_t lo, cnt; range(g, n, m, &lo, &cnt);
            if (!cnt) return;
            bool ok = hipSetDevice(g->dev[m]) == hipSuccess;
            for (size_t sgm = 0; sgm < segs && ok; ++sgm) {
                u64* d = shards[m] + sgm * cnt * ew;
                const size_t hoff = (sgm * n + lo) * ew;
                ok = (to_device ? hipMemcpyAsync(d, host_c + hoff, cnt * ew * 8, hipMemcpyHostToDevice, g->ctx[m]->stream)
                                : hipMemcpyAsync(host_m + hoff, d, cnt * ew * 8, hipMemcpyDeviceToHost, g->ctx[m]->stream)) == hipSuccess;
            }
            ok = ok && hipStreamSynchronize(g->ctx[m]->stream) == hipSuccess;
            if (!ok) { (void)hipGetLastError(); bad[m] = 1; }
        });
        for (int m = 0; m < g->G; ++m) if (bad[m]) { gset_err(g, "member " + std::to_string(m) + ": host transfer failed"); return ARKMPC_ERR_HIP; }
        return ARKMPC_OK;
    }
    for (int m = 0; m < g->G; ++m) {
        size_t lo, cnt; range(g, n, m, &lo, &cnt);
        if (!cnt) continue;
        GHIP(g, hipSetDevice(g->dev[m]));
        for (size_t s = 0; s < segs; ++s) {
            u64* d = shards[m] + s * cnt * ew;
            const size_t hoff = (s * n + lo) * ew;
            if (to_device) GHIP(g, hipMemcpyAsync(d, host_c + hoff, cnt * ew * 8, hipMemcpyHostToDevice, g->ctx[m]->stream));
            else GHIP(g, hipMemcpyAsync(host_m + hoff, d, cnt * ew * 8, hipMemcpyDeviceToHost, g->ctx[m]->stream));
        }
    }
    for (int m = 0; m < g->G; ++m) { GHIP(g, hipSetDevice(g->dev[m])); GHIP(g, hipStreamSynchronize(g->ctx[m]->stream)); }
    return ARKMPC_OK;                                      // (the pin ends here: every copy has completed)
}
int arkmpc_group_scatter_h2d(arkmpc_group* g, size_t n, size_t segs, size_t elem_words, const uint64_t* host, uint64_t* const* shards) {
    return host_xfer(g, true, n, segs, elem_words, host, nullptr, shards);
}
int arkmpc_group_gather_d2h(arkmpc_group* g, size_t n, size_t segs, size_t elem_words, const uint64_t* const* shards, uint64_t* host) {
    return host_xfer(g, false, n, segs, elem_words, nullptr, host, (u64* const*)shards);
}

// Vec<ScalarShare> (arkworks records on the host) <-> sharded ScalarShare vector in `layout`.  Split columns: the vector is pinned once and
// every member's import kernel reads ITS range of the records in place over its own link and writes the two columns (ark_import_split):
// no staging block, no split pass, all members in flight together.  A vector that cannot be addressed in place (not pinnable, or only
// 8-byte aligned) goes the old way per member: staging block, copy, split.
int arkmpc_group_shares_from_host(arkmpc_group* g, int layout, size_t n, const uint64_t* host_records, uint64_t* const* shards) {
    if (!g) return ARKMPC_ERR_BAD_ARG;
    if (!layout_ok(layout)) return gbad(g, "bad layout");
    if (layout == ARKMPC_LAYOUT_AOS) return host_xfer(g, true, n, 1, 8, host_records, nullptr, shards);
    if (n && !host_records) return gbad(g, "null host pointer");
    int rc = shards_ok(g, n, (const void* const*)shards, "null shard pointer");
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(g->mu);
    HostPins pins;
    Place src;
    if (n) {
        src = classify_and_hold(g->ctx[0], pins, host_records, n * 64);
        if (src.kind == Mem::Pageable) { pins.pin(host_records, n * 64); src = classify(g->ctx[0], host_records, n * 64); }
    }
    const bool in_place = src.kind == Mem::Pinned && src.zc();
    std::vector<void*> tmp(g->G, nullptr);
    for (int m = 0; m < g->G && rc == ARKMPC_OK; ++m) {
        size_t lo, cnt; range(g, n, m, &lo, &cnt);
        if (!cnt) continue;
        if (hipSetDevice(g->dev[m]) != hipSuccess) { gset_err(g, "hipSetDevice failed"); rc = ARKMPC_ERR_HIP; break; }
        if (in_place) {
            // the mapped alias of a pinned range is valid on every device of the node (one registration, portable across the GPUs of the process)
            rc = ark_import_split(g->ctx[m]->stream, cnt, (const char*)src.dev + 64 * lo, shards[m], shards[m] + 4 * cnt);
            if (rc) gset_err(g, "import kernel launch failed");
            continue;
        }
        rc = arkmpc_malloc(g->ctx[m], cnt * 64, &tmp[m]);
        if (rc) { gfail(g, m, rc, "arkmpc_malloc"); break; }
        if (hipMemcpyAsync(tmp[m], host_records + 8 * lo, cnt * 64, hipMemcpyHostToDevice, g->ctx[m]->stream) != hipSuccess) { gset_err(g, "H2D failed"); rc = ARKMPC_ERR_HIP; break; }
        rc = arkmpc_share_split(g->ctx[m], cnt, (const u64*)tmp[m], shards[m], shards[m] + 4 * cnt);
        if (rc) gfail(g, m, rc, "arkmpc_share_split");
    }
    for (int m = 0; m < g->G; ++m) {
        int r = arkmpc_sync(g->ctx[m]);                       // the host vector may go away when we return
        if (r && !rc) rc = gfail(g, m, r, "arkmpc_sync");
        if (tmp[m]) (void)arkmpc_free(g->ctx[m], tmp[m]);
    }
    return rc;
}
int arkmpc_group_shares_to_host(arkmpc_group* g, int layout, size_t n, const uint64_t* const* shards, uint64_t* host_records) {
    if (!g) return ARKMPC_ERR_BAD_ARG;
    if (!layout_ok(layout)) return gbad(g, "bad layout");
    if (layout == ARKMPC_LAYOUT_AOS) return host_xfer(g, false, n, 1, 8, nullptr, host_records, (u64* const*)shards);
    if (n && !host_records) return gbad(g, "null host pointer");
    int rc = shards_ok(g, n, (const void* const*)shards, "null shard pointer");
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(g->mu);
    HostPins pins;
    if (n) pins.pin(host_records, n * 64);
    std::vector<void*> tmp(g->G, nullptr);
    for (int m = 0; m < g->G && rc == ARKMPC_OK; ++m) {
        size_t lo, cnt; range(g, n, m, &lo, &cnt);
        if (!cnt) continue;
        rc = arkmpc_malloc(g->ctx[m], cnt * 64, &tmp[m]);
        if (rc) { gfail(g, m, rc, "arkmpc_malloc"); break; }
        rc = arkmpc_share_join(g->ctx[m], cnt, shards[m], shards[m] + 4 * cnt, (u64*)tmp[m]);
        if (rc) { gfail(g, m, rc, "arkmpc_share_join"); break; }
        if (hipSetDevice(g->dev[m]) != hipSuccess ||
            hipMemcpyAsync(host_records + 8 * lo, tmp[m], cnt * 64, hipMemcpyDeviceToHost, g->ctx[m]->stream) != hipSuccess) { gset_err(g, "D2H failed"); rc = ARKMPC_ERR_HIP; }
    }
    for (int m = 0; m < g->G; ++m) {
        int r = arkmpc_sync(g->ctx[m]);
        if (r && !rc) rc = gfail(g, m, r, "arkmpc_sync");
        if (tmp[m]) (void)arkmpc_free(g->ctx[m], tmp[m]);
    }
    return rc;
}

// ---- streaming sessions over the group: host vectors in, host vectors out, one PCIe link per member ----------------------------------
// AuthenticatedScalarResult::batch_mul (authenticated_scalar.rs:848-879) for a party whose operands ARE host memory (what
// benches/batch_ops.rs:19-39 times) and that owns several GPUs.  Host-fed, the path is bound by the host link, 20x below the kernels'
// rate, so the links -- one per GPU -- are what more GPUs add.  A group session is one range session per member
// (arkmpc_hostmul_begin_range on [g n/G, (g+1) n/G) of the SAME host vectors: x + 8 lo, ..., d at out_de + 4 lo, e at out_de + 4 (n + lo)),
// each on its member's context, device and link.  The caller's vectors are pinned once per call, whole (nothing to do for vectors it
// registered or allocated pinned); the member sessions then run their phases on their ranges -- as kernels that address the ranges in place when the
// CALLER holds the vectors in pinned memory, through the DMA pipeline when the pin is the library's own (no kernel addresses a vector the
// library registered itself: DESIGN section 4) -- and because every member call only ENQUEUES (begin_range, finish_async), one host thread keeps
// all G links busy at once; _finish ends the members
// after all of them have been started.  Vectors that cannot be pinned travel as the runtime's pageable copies, member after member.
struct arkmpc_group_hostmul {
    arkmpc_group* g = nullptr;
    size_t n = 0;
    std::vector<arkmpc_hostmul*> ses;                     // one per member (null: empty range)
    HostPins pins_in, pins_de, pins_c, pins_peer, pins_out;
};
namespace {
// ends every member session that is still open (their streams drain); keeps the first error
int ghm_end_all(arkmpc_group_hostmul* s, int rc) {
    arkmpc_group* g = s->g;
    for (int m = 0; m < g->G; ++m) {
        if (!s->ses[m]) continue;
        const int r = arkmpc_hostmul_end(s->ses[m]);
        s->ses[m] = nullptr;
        if (r && !rc) rc = gfail(g, m, r, "arkmpc_hostmul_end");
    }
    delete s;                                             // (the whole-vector pins end with it: nothing is in flight any more)
    return rc;
}
}  // namespace
int arkmpc_group_hostmul_begin(arkmpc_group* g, size_t n, const uint64_t* x, const uint64_t* y, const uint64_t* a, const uint64_t* b, const uint64_t* c,
                               uint64_t* out_de, arkmpc_group_hostmul** out_session) {
    if (!g) return ARKMPC_ERR_BAD_ARG;
    if (!out_session) return gbad(g, "null out_session");
    *out_session = nullptr;
    if (n && (!x || !y || !a || !b || !c || !out_de)) return gbad(g, "null pointer");
    if (n > ((size_t)1 << 40)) return gbad(g, "batch too large");
    std::lock_guard<std::mutex> lk(g->mu);
    arkmpc_group_hostmul* s = new arkmpc_group_hostmul();
    s->g = g; s->n = n;
    s->ses.assign(g->G, nullptr);
    const size_t rec = n * 64;
    if (n) { s->pins_in.pin(x, rec); s->pins_in.pin(y, rec); s->pins_in.pin(a, rec); s->pins_in.pin(b, rec); s->pins_de.pin(out_de, rec); }
    // a phase whose vectors are pageable is made of blocking copies: one thread per member then (members_in_threads)
    const bool thr = n && members_in_threads(g) && (pageable_host(g, x, rec) || pageable_host(g, y, rec) || pageable_host(g, a, rec) || pageable_host(g, b, rec) ||
                                                     pageable_host(g, out_de, rec));
    std::vector<int> rcs(g->G, ARKMPC_OK);
    for_members(g, thr, [&](int m) {
        size_t lo, cnt; range(g, n, m, &lo, &cnt);
        if (!cnt) return;
        rcs[m] = arkmpc_hostmul_begin_range(g->ctx[m], cnt, x + 8 * lo, y + 8 * lo, a + 8 * lo, b + 8 * lo, c + 8 * lo, out_de + 4 * lo, out_de + 4 * (n + lo),
                                            ARKMPC_HOSTMUL_NO_PIN, &s->ses[m]);
    });
    for (int m = 0; m < g->G; ++m) if (rcs[m]) return ghm_end_all(s, gfail(g, m, rcs[m], "arkmpc_hostmul_begin_range"));
    if (n) s->pins_c.pin(c, rec);                          // under the members' phase 1: c is not looked at before _wait_de / _finish
    *out_session = s;
    return ARKMPC_OK;
}
// leading gates of d AND e that have landed in out_de: the members' ranges in order, up to and including the first incomplete one's progress
int arkmpc_group_hostmul_poll_de(arkmpc_group_hostmul* s, size_t* out_gates) {
    if (!s || !out_gates) return ARKMPC_ERR_BAD_ARG;
    arkmpc_group* g = s->g;
    size_t done = 0;
    for (int m = 0; m < g->G; ++m) {
        size_t lo, cnt; range(g, s->n, m, &lo, &cnt);
        if (!cnt) continue;
        size_t k = 0;
        if (arkmpc_hostmul_poll_de(s->ses[m], &k) != ARKMPC_OK) break;
        done = lo + k;
        if (k < cnt) break;
    }
    *out_gates = done;
    return ARKMPC_OK;
}
int arkmpc_group_hostmul_wait_de(arkmpc_group_hostmul* s) {
    if (!s) return ARKMPC_ERR_BAD_ARG;
    arkmpc_group* g = s->g;
    std::lock_guard<std::mutex> lk(g->mu);
    for (int m = 0; m < g->G; ++m) if (s->ses[m]) GCALL(g, m, arkmpc_hostmul_wait_de(s->ses[m]));
    s->pins_de.release();                                  // the payload has been produced: the vector is the caller's again
    return ARKMPC_OK;
}
int arkmpc_group_hostmul_finish(arkmpc_group_hostmul* s, int party_id, const uint64_t mac_key[4], const uint64_t* peer_de, uint64_t* out) {
    if (!s) return ARKMPC_ERR_BAD_ARG;
    arkmpc_group* g = s->g;
    std::lock_guard<std::mutex> lk(g->mu);
    const size_t n = s->n;
    int rc = ARKMPC_OK;
    if (party_id != 0 && party_id != 1) rc = gbad(g, "party_id must be 0 or 1");
    else if (!mac_key) rc = gbad(g, "null mac_key");
    else if (n && (!peer_de || !out)) rc = gbad(g, "null pointer");
    if (!rc && n) { s->pins_peer.pin(peer_de, n * 64); s->pins_out.pin(out, n * 64); }
    if (!rc) {                                             // phase 2 of every member enqueued (or, for pageable vectors, copied by its own thread) before any of them is waited for
        const bool thr = n && members_in_threads(g) && (pageable_host(g, peer_de, n * 64) || pageable_host(g, out, n * 64));
        std::vector<int> rcs(g->G, ARKMPC_OK);
        for_members(g, thr, [&](int m) {
            size_t lo, cnt; range(g, n, m, &lo, &cnt);
            if (!cnt) return;
            rcs[m] = arkmpc_hostmul_finish_async(s->ses[m], party_id, mac_key, peer_de + 4 * lo, peer_de + 4 * (n + lo), out + 8 * lo);
        });
        for (int m = 0; m < g->G; ++m) if (rcs[m] && !rc) rc = gfail(g, m, rcs[m], "arkmpc_hostmul_finish_async");
    }
    return ghm_end_all(s, rc);
}
int arkmpc_group_hostmul_abort(arkmpc_group_hostmul* s) {
    if (!s) return ARKMPC_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lk(s->g->mu);
    return ghm_end_all(s, ARKMPC_OK);
}

// ---- device-to-device movement: direct peer writes -------------------------------------------------------------------------------
// every member pushes its `segs` segments to their place in the full buffer on `root`; root's stream then waits (on the device) for
// all pushes.  Asynchronous: consumers enqueue on root's stream, or call arkmpc_group_sync.
int arkmpc_group_gather(arkmpc_group* g, size_t n, size_t segs, size_t elem_words, const uint64_t* const* shards, int root, uint64_t* out_on_root) {
    if (!g) return ARKMPC_ERR_BAD_ARG;
    if (root < 0 || root >= g->G || !segs || !elem_words) return gbad(g, "group_gather: bad arguments");
    if (n && !out_on_root) return gbad(g, "null output");
    int rc = shards_ok(g, n, (const void* const*)shards, "null shard pointer");
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(g->mu);
    if ((rc = sources_wait_for(g, root))) return rc;
    for (int m = 0; m < g->G; ++m) {
        size_t lo, cnt; range(g, n, m, &lo, &cnt);
        for (size_t s = 0; s < segs && cnt; ++s) {
            rc = push(g, m, root, out_on_root + (s * n + lo) * elem_words, shards[m] + s * cnt * elem_words, cnt * elem_words * 8);
            if (rc) return rc;
        }
        if (m != root) { GHIP(g, hipSetDevice(g->dev[m])); GHIP(g, hipEventRecord(g->ev[m], g->ctx[m]->stream)); }
    }
    GHIP(g, hipSetDevice(g->dev[root]));
    for (int m = 0; m < g->G; ++m) if (m != root) GHIP(g, hipStreamWaitEvent(g->ctx[root]->stream, g->ev[m], 0));
    return ARKMPC_OK;
}
// all-gather: every member pushes its range into EVERY member's full buffer -- G*(G-1) point-to-point copies, one per directed link
// of the xGMI mesh, all in flight together.  The pushes to the different destinations are issued round-robin (member m starts with
// destination m+1) so that no destination's links are hit by all sources in the same instant.
int arkmpc_group_allgather(arkmpc_group* g, size_t n, size_t segs, size_t elem_words, const uint64_t* const* shards, uint64_t* const* outs) {
    if (!g) return ARKMPC_ERR_BAD_ARG;
    if (!segs || !elem_words || !outs) return gbad(g, "group_allgather: bad arguments");
    int rc = shards_ok(g, n, (const void* const*)shards, "null shard pointer");
    if (rc) return rc;
    for (int m = 0; m < g->G; ++m) if (n && !outs[m]) return gbad(g, "null output");
    std::lock_guard<std::mutex> lk(g->mu);
    for (int to = 0; to < g->G; ++to) if ((rc = sources_wait_for(g, to))) return rc;
    for (int step = 0; step < g->G; ++step)
        for (int m = 0; m < g->G; ++m) {
            const int to = (m + step) % g->G;
            size_t lo, cnt; range(g, n, m, &lo, &cnt);
            for (size_t s = 0; s < segs && cnt; ++s) {
                rc = push(g, m, to, outs[to] + (s * n + lo) * elem_words, shards[m] + s * cnt * elem_words, cnt * elem_words * 8);
                if (rc) return rc;
            }
        }
    for (int m = 0; m < g->G; ++m) { GHIP(g, hipSetDevice(g->dev[m])); GHIP(g, hipEventRecord(g->ev[m], g->ctx[m]->stream)); }
    for (int to = 0; to < g->G; ++to) {
        GHIP(g, hipSetDevice(g->dev[to]));
        for (int m = 0; m < g->G; ++m) if (m != to) GHIP(g, hipStreamWaitEvent(g->ctx[to]->stream, g->ev[m], 0));
    }
    return ARKMPC_OK;
}
// the reverse of gather: a full buffer on `root` (e.g. the peer party's d||e as it arrived) cut into the members' shards.  Root's
// stream pushes every range over a different link; each member's stream waits for its own range only.
int arkmpc_group_scatter(arkmpc_group* g, size_t n, size_t segs, size_t elem_words, const uint64_t* src_on_root, int root, uint64_t* const* shards) {
    if (!g) return ARKMPC_ERR_BAD_ARG;
    if (root < 0 || root >= g->G || !segs || !elem_words) return gbad(g, "group_scatter: bad arguments");
    if (n && !src_on_root) return gbad(g, "null source");
    int rc = shards_ok(g, n, (const void* const*)shards, "null shard pointer");
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(g->mu);
    // the members must have finished with their shard buffers before root overwrites them: root waits for each member's stream
    for (int m = 0; m < g->G; ++m) if (m != root) { GHIP(g, hipSetDevice(g->dev[m])); GHIP(g, hipEventRecord(g->ev[m], g->ctx[m]->stream)); }
    GHIP(g, hipSetDevice(g->dev[root]));
    for (int m = 0; m < g->G; ++m) if (m != root) GHIP(g, hipStreamWaitEvent(g->ctx[root]->stream, g->ev[m], 0));
    for (int m = 0; m < g->G; ++m) {
        size_t lo, cnt; range(g, n, m, &lo, &cnt);
        for (size_t s = 0; s < segs && cnt; ++s) {
            rc = push(g, root, m, shards[m] + s * cnt * elem_words, src_on_root + (s * n + lo) * elem_words, cnt * elem_words * 8);
            if (rc) return rc;
        }
    }
    GHIP(g, hipSetDevice(g->dev[root]));
    GHIP(g, hipEventRecord(g->ev[root], g->ctx[root]->stream));
    for (int m = 0; m < g->G; ++m) if (m != root) { GHIP(g, hipSetDevice(g->dev[m])); GHIP(g, hipStreamWaitEvent(g->ctx[m]->stream, g->ev[root], 0)); }
    return ARKMPC_OK;
}

// ---- the path, range-sharded -----------------------------------------------------------------------------------------------------
static int group_mask(arkmpc_group* g, int layout, size_t n, const uint64_t* const* x, const uint64_t* const* y, const uint64_t* const* a,
                      const uint64_t* const* b, uint64_t* const* out_de, int root, uint64_t* out_on_root) {
    if (!g) return ARKMPC_ERR_BAD_ARG;
    if (!layout_ok(layout)) return gbad(g, "bad layout");
    int rc = shards_ok(g, n, (const void* const*)x, "null x shard");
    if (!rc) rc = shards_ok(g, n, (const void* const*)y, "null y shard");
    if (!rc) rc = shards_ok(g, n, (const void* const*)a, "null a shard");
    if (!rc) rc = shards_ok(g, n, (const void* const*)b, "null b shard");
    if (!rc && !out_on_root) rc = shards_ok(g, n, (const void* const*)out_de, "null d||e shard");
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(g->mu);
    if (out_on_root && (rc = sources_wait_for(g, root))) return rc;       // the kernels below store into root's buffer
    for (int m = 0; m < g->G; ++m) {
        size_t lo, cnt; range(g, n, m, &lo, &cnt);
        if (!cnt) continue;
        const ShareView vx = share_view(layout, x[m], cnt), vy = share_view(layout, y[m], cnt), va = share_view(layout, a[m], cnt), vb = share_view(layout, b[m], cnt);
        u64* od = out_on_root ? out_on_root + 4 * lo : out_de[m];
        u64* oe = out_on_root ? out_on_root + 4 * (n + lo) : out_de[m] + 4 * cnt;
        GCALL(g, m, arkmpc_beaver_mask_to(g->ctx[m], cnt, vx.s, vx.stride, vy.s, vy.stride, va.s, va.stride, vb.s, vb.stride, od, oe));
    }
    if (out_on_root) {                         // root's stream sees the full buffer once every member's kernel has stored its range
        for (int m = 0; m < g->G; ++m) if (m != root) { GHIP(g, hipSetDevice(g->dev[m])); GHIP(g, hipEventRecord(g->ev[m], g->ctx[m]->stream)); }
        GHIP(g, hipSetDevice(g->dev[root]));
        for (int m = 0; m < g->G; ++m) if (m != root) GHIP(g, hipStreamWaitEvent(g->ctx[root]->stream, g->ev[m], 0));
    }
    return ARKMPC_OK;
}
int arkmpc_group_beaver_mask(arkmpc_group* g, int layout, size_t n, const uint64_t* const* x, const uint64_t* const* y, const uint64_t* const* a,
                             const uint64_t* const* b, uint64_t* const* out_de) {
    return group_mask(g, layout, n, x, y, a, b, out_de, -1, nullptr);
}
// K1 whose stores ARE the gather: member m's kernel writes d and e of its range straight into the full d||e buffer on `root` through
// the peer mapping (no staging shard, no copy pass).  Needs peer access from every member to root.
int arkmpc_group_beaver_mask_gathered(arkmpc_group* g, int layout, size_t n, const uint64_t* const* x, const uint64_t* const* y,
                                      const uint64_t* const* a, const uint64_t* const* b, int root, uint64_t* out_de_on_root) {
    if (!g) return ARKMPC_ERR_BAD_ARG;
    if (root < 0 || root >= g->G) return gbad(g, "bad root member");
    if (n && !out_de_on_root) return gbad(g, "null output");
    if ((uintptr_t)out_de_on_root & 15) return gbad(g, "output not 16-byte aligned");
    for (int m = 0; m < g->G; ++m) if (!g->peer[(size_t)m * g->G + root]) { gset_err(g, "no peer access to the root device: use arkmpc_group_beaver_mask + arkmpc_group_gather"); return ARKMPC_ERR_UNSUPPORTED; }
    return group_mask(g, layout, n, x, y, a, b, nullptr, root, out_de_on_root);
}
int arkmpc_group_beaver_finish_fused(arkmpc_group* g, int layout, size_t n, int party_id, const uint64_t mac_key[4], const uint64_t* const* my_de,
                                     const uint64_t* const* peer_de, const uint64_t* const* a, const uint64_t* const* b, const uint64_t* const* c,
                                     uint64_t* const* out) {
    if (!g) return ARKMPC_ERR_BAD_ARG;
    if (!layout_ok(layout)) return gbad(g, "bad layout");
    if (!mac_key) return gbad(g, "null mac_key");
    int rc = shards_ok(g, n, (const void* const*)my_de, "null d||e shard");
    if (!rc) rc = shards_ok(g, n, (const void* const*)peer_de, "null peer d||e shard");
    if (!rc) rc = shards_ok(g, n, (const void* const*)a, "null a shard");
    if (!rc) rc = shards_ok(g, n, (const void* const*)b, "null b shard");
    if (!rc) rc = shards_ok(g, n, (const void* const*)c, "null c shard");
    if (!rc) rc = shards_ok(g, n, (const void* const*)out, "null output shard");
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(g->mu);
    for (int m = 0; m < g->G; ++m) {
        size_t lo, cnt; range(g, n, m, &lo, &cnt);
        if (!cnt) continue;
        const ShareView va = share_view(layout, a[m], cnt), vb = share_view(layout, b[m], cnt), vc = share_view(layout, c[m], cnt);
        const ShareView vo = share_view(layout, out[m], cnt);
        GCALL(g, m, arkmpc_beaver_finish_fused_from(g->ctx[m], cnt, party_id, mac_key, my_de[m], my_de[m] + 4 * cnt, peer_de[m], peer_de[m] + 4 * cnt,
                                                    va.s, va.m, va.stride, vb.s, vb.m, vb.stride, vc.s, vc.m, vc.stride, (u64*)vo.s, (u64*)vo.m, vo.stride));
    }
    return ARKMPC_OK;
}
// the `.share()` projection a party sends in open_batch (authenticated_scalar.rs:141-145): AoS records -> Scalars per range; with
// split columns the payload IS the share column (a device copy keeps the interface uniform)
int arkmpc_group_share_extract(arkmpc_group* g, int layout, size_t n, const uint64_t* const* shares, uint64_t* const* out_values) {
    if (!g) return ARKMPC_ERR_BAD_ARG;
    if (!layout_ok(layout)) return gbad(g, "bad layout");
    int rc = shards_ok(g, n, (const void* const*)shares, "null share shard");
    if (!rc) rc = shards_ok(g, n, (const void* const*)out_values, "null output shard");
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(g->mu);
    for (int m = 0; m < g->G; ++m) {
        size_t lo, cnt; range(g, n, m, &lo, &cnt);
        if (!cnt) continue;
        if (layout == ARKMPC_LAYOUT_SPLIT) GCALL(g, m, arkmpc_memcpy_d2d(g->ctx[m], out_values[m], shares[m], cnt * 32));
        else GCALL(g, m, arkmpc_share_extract(g->ctx[m], cnt, shares[m], out_values[m]));
    }
    return ARKMPC_OK;
}
// K2+K4 per range (authenticated_scalar.rs:161-171, :299-311)
int arkmpc_group_open_and_mac_check(arkmpc_group* g, int layout, size_t n, const uint64_t mac_key[4], const uint64_t* const* shares,
                                    const uint64_t* const* peer_values, uint64_t* const* out_opened, uint64_t* const* out_chk) {
    if (!g) return ARKMPC_ERR_BAD_ARG;
    if (!layout_ok(layout)) return gbad(g, "bad layout");
    if (!mac_key) return gbad(g, "null mac_key");
    int rc = shards_ok(g, n, (const void* const*)shares, "null share shard");
    if (!rc) rc = shards_ok(g, n, (const void* const*)peer_values, "null peer shard");
    if (!rc) rc = shards_ok(g, n, (const void* const*)out_opened, "null opened shard");
    if (!rc) rc = shards_ok(g, n, (const void* const*)out_chk, "null chk shard");
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(g->mu);
    for (int m = 0; m < g->G; ++m) {
        size_t lo, cnt; range(g, n, m, &lo, &cnt);
        if (!cnt) continue;
        const ShareView v = share_view(layout, shares[m], cnt);
        GCALL(g, m, arkmpc_open_and_mac_check_v(g->ctx[m], cnt, mac_key, v.s, v.m, v.stride, peer_values[m], out_opened[m], out_chk[m]));
    }
    return ARKMPC_OK;
}
// K5 per range, ONE synchronisation per member, AND of the flags (authenticated_scalar.rs:218-219)
int arkmpc_group_mac_verify(arkmpc_group* g, size_t n, const uint64_t* const* mine, const uint64_t* const* peer, int* out_ok) {
    if (!g) return ARKMPC_ERR_BAD_ARG;
    if (!out_ok) return gbad(g, "null out_ok");
    int rc = shards_ok(g, n, (const void* const*)mine, "null shard");
    if (!rc) rc = shards_ok(g, n, (const void* const*)peer, "null peer shard");
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(g->mu);
    for (int m = 0; m < g->G; ++m) {
        size_t lo, cnt; range(g, n, m, &lo, &cnt);
        if (cnt) GCALL(g, m, arkmpc_mac_verify_async(g->ctx[m], cnt, mine[m], peer[m]));
    }
    int all = 1;
    for (int m = 0; m < g->G; ++m) {             // every member's flag is collected (and cleared) even after the first failure
        int ok = 0;
        GCALL(g, m, arkmpc_mac_verify_result(g->ctx[m], &ok));
        all &= ok;
    }
    *out_ok = all;
    return ARKMPC_OK;
}

// H1 over a sharded vector (commitment.rs:63-89): the message is BE(v_0) || ... || BE(v_{n-1}) || BE(blinder) in index order, i.e.
// member 0's range, then member 1's, ...  Each member converts its range with K6 and copies it to its own pinned slots (two per
// member, all members prefetching concurrently over their own PCIe links); the host sponge walks the slots in order.  No device
// gather is needed for a commitment.
int arkmpc_group_commit_sha3(arkmpc_group* g, size_t n, const uint64_t* const* values, const uint64_t blinder[4], uint64_t out_commitment[4]) {
    if (!g) return ARKMPC_ERR_BAD_ARG;
    if (!blinder || !out_commitment) return gbad(g, "null blinder/out");
    int rc = shards_ok(g, n, (const void* const*)values, "null value shard");
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(g->mu);
    const size_t CH = kCommitChunk;
    for (int m = 0; m < g->G; ++m) {
        if (g->pin[2 * m]) continue;
        GHIP(g, hipSetDevice(g->dev[m]));
        for (int s = 0; s < 2; ++s) {
            GHIP(g, hipHostMalloc((void**)&g->pin[2 * m + s], CH * 32));
            void* p = nullptr;
            GCALL(g, m, arkmpc_malloc(g->ctx[m], CH * 32, &p));
            g->dstage[2 * m + s] = (unsigned char*)p;
            GHIP(g, hipEventCreateWithFlags(&g->cev[2 * m + s], hipEventDisableTiming));
        }
    }
    auto issue = [&](int m, size_t c, size_t cnt_m) -> int {
        const size_t off = c * CH, cnt = (cnt_m - off < CH) ? (cnt_m - off) : CH;
        const int s = (int)(c & 1);
        GCALL(g, m, arkmpc_scalar_to_bytes_be(g->ctx[m], cnt, values[m] + 4 * off, g->dstage[2 * m + s]));
        GHIP(g, hipSetDevice(g->dev[m]));
        GHIP(g, hipMemcpyAsync(g->pin[2 * m + s], g->dstage[2 * m + s], cnt * 32, hipMemcpyDeviceToHost, g->ctx[m]->stream));
        GHIP(g, hipEventRecord(g->cev[2 * m + s], g->ctx[m]->stream));
        return ARKMPC_OK;
    };
    std::vector<size_t> cnts(g->G), nch(g->G);
    for (int m = 0; m < g->G; ++m) {
        size_t lo; range(g, n, m, &lo, &cnts[m]);
        nch[m] = (cnts[m] + CH - 1) / CH;
        for (size_t c = 0; c < nch[m] && c < 2; ++c) { rc = issue(m, c, cnts[m]); if (rc) return rc; }     // prefetch: both slots of every member
    }
    Sha3State sh;
    sha3_256_init(&sh);
    for (int m = 0; m < g->G; ++m)
        for (size_t c = 0; c < nch[m]; ++c) {
            const int s = (int)(c & 1);
            GHIP(g, hipEventSynchronize(g->cev[2 * m + s]));
            const size_t off = c * CH, cnt = (cnts[m] - off < CH) ? (cnts[m] - off) : CH;
            sha3_256_update(&sh, g->pin[2 * m + s], cnt * 32);
            if (c + 2 < nch[m]) { rc = issue(m, c + 2, cnts[m]); if (rc) return rc; }                       // the slot just absorbed is free again
        }
    unsigned char be[32], dig[32];
    host_to_bytes_be(g->field_id, blinder, be);
    sha3_256_update(&sh, be, 32);
    sha3_256_final(&sh, dig);
    host_from_be_bytes_mod_order(g->field_id, dig, out_commitment);
    return ARKMPC_OK;
}

// Variable-base MSM over sharded (point, scalar) pairs: every member folds its range with the bucket method (arkmpc_g1_msm), the G
// partial results are added on member 0 (CurvePoint::msm, curve.rs:549-560; the cross-GPU point reduction of SURVEY.md section 8f-3).
// The per-member calls block on their Horner tails, so they run on one host thread each.  BN254 contexts only.
}  // extern "C"
namespace {
typedef int (*MsmFn)(arkmpc_ctx*, size_t, const uint64_t*, const uint64_t*, uint64_t*);
typedef int (*SumFn)(arkmpc_ctx*, size_t, const uint64_t*, uint64_t*);
// one bucket MSM per member over its range, the G partial points pushed to member 0 and added there; pw = u64 words of a point
int group_msm(arkmpc_group* g, size_t n, const uint64_t* const* points, const uint64_t* const* scalars, uint64_t* out_point, size_t pw, int field, MsmFn msm, SumFn sum_fn,
              const char* what) {
    if (!g) return ARKMPC_ERR_BAD_ARG;
    if (!out_point) return gbad(g, "null output");
    if (g->field_id != field) { gset_err(g, what); return ARKMPC_ERR_UNSUPPORTED; }
    int rc = shards_ok(g, n, (const void* const*)points, "null point shard");
    if (!rc) rc = shards_ok(g, n, (const void* const*)scalars, "null scalar shard");
    if (rc) return rc;
    const size_t pb = pw * 8;
    std::lock_guard<std::mutex> lk(g->mu);
    std::vector<void*> part(g->G, nullptr);
    std::vector<int> rcs(g->G, ARKMPC_OK);
    for (int m = 0; m < g->G; ++m) { rc = arkmpc_malloc(g->ctx[m], pb, &part[m]); if (rc) { gfail(g, m, rc, "arkmpc_malloc"); break; } }
    if (!rc) {
        std::vector<std::thread> th;
        for (int m = 0; m < g->G; ++m)
            th.emplace_back([&, m] {
                size_t lo, cnt; range(g, n, m, &lo, &cnt);
                rcs[m] = msm(g->ctx[m], cnt, points[m], scalars[m], (u64*)part[m]);
                if (rcs[m] == ARKMPC_OK) rcs[m] = arkmpc_sync(g->ctx[m]);
            });
        for (auto& t : th) t.join();
        for (int m = 0; m < g->G && !rc; ++m) if (rcs[m]) rc = gfail(g, m, rcs[m], "member MSM");
    }
    void* all = nullptr; void* sum = nullptr;
    if (!rc) { rc = arkmpc_malloc(g->ctx[0], (size_t)g->G * pb, &all); if (rc) gfail(g, 0, rc, "arkmpc_malloc"); }
    if (!rc) { rc = arkmpc_malloc(g->ctx[0], pb, &sum); if (rc) gfail(g, 0, rc, "arkmpc_malloc"); }
    for (int m = 0; m < g->G && !rc; ++m) rc = push(g, m, 0, (char*)all + (size_t)m * pb, part[m], pb);
    for (int m = 1; m < g->G && !rc; ++m) {
        if (hipSetDevice(g->dev[m]) != hipSuccess || hipEventRecord(g->ev[m], g->ctx[m]->stream) != hipSuccess ||
            hipSetDevice(g->dev[0]) != hipSuccess || hipStreamWaitEvent(g->ctx[0]->stream, g->ev[m], 0) != hipSuccess) { gset_err(g, "event ordering failed"); rc = ARKMPC_ERR_HIP; }
    }
    if (!rc) { rc = sum_fn(g->ctx[0], (size_t)g->G, (const u64*)all, (u64*)sum); if (rc) gfail(g, 0, rc, "point sum"); }
    if (!rc) { rc = arkmpc_memcpy_d2h(g->ctx[0], out_point, sum, pb); if (rc) gfail(g, 0, rc, "arkmpc_memcpy_d2h"); }
    for (int m = 0; m < g->G; ++m) if (part[m]) (void)arkmpc_free(g->ctx[m], part[m]);
    if (all) (void)arkmpc_free(g->ctx[0], all);
    if (sum) (void)arkmpc_free(g->ctx[0], sum);
    return rc;
}
}  // namespace
extern "C" {
int arkmpc_group_g1_msm(arkmpc_group* g, size_t n, const uint64_t* const* points, const uint64_t* const* scalars, uint64_t out_point[12]) {
    return group_msm(g, n, points, scalars, out_point, 12, ARKMPC_BN254_FR, arkmpc_g1_msm, arkmpc_g1_sum, "group MSM over G1 needs a BN254 Fr group");
}
int arkmpc_group_ed_msm(arkmpc_group* g, size_t n, const uint64_t* const* points, const uint64_t* const* scalars, uint64_t out_point[16]) {
    return group_msm(g, n, points, scalars, out_point, 16, ARKMPC_CURVE25519_FR, arkmpc_ed_msm, arkmpc_ed_sum, "group MSM over Curve25519 needs a CURVE25519_FR group");
}

}  // extern "C"
