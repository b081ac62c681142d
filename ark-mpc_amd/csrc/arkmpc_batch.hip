// arkmpc_batch.hip -- the device-batch carrier: an opaque, reference-counted handle to ONE batch value resident in HBM.
//
// Why: in the reference every value that flows from gate to gate is a `ResultValue<C>` (fabric/result.rs:47-64) -- a host enum
// whose batch variants own a `Vec<Scalar>` / `Vec<CurvePoint>`.  A patched fabric that keeps gate outputs on the GPU needs a
// variant `ResultValue::DeviceBatch(ArkBatch)` whose payload is THIS handle: element kind (which enum variant it stands for),
// element count, layout tag (arkworks AoS records, or the engine-native split columns for ScalarShares) and the device
// storage.  Handles are what the batch-level entry points at the bottom take, so a chain of gates (share -> mul -> mul -> open)
// never round-trips through host vectors; `arkmpc_batch_to_host` materialises the arkworks `Vec<T>` only where the caller
// awaits a value.  Slices (`&v[lo..hi]`, authenticated_scalar.rs batch ops over sub-ranges, sharding by index range) are views
// that share the parent's storage and keep it alive.
#include "arkmpc_internal.hpp"
#include <atomic>
#include <cstdlib>
#include <initializer_list>

struct arkmpc_batch {
    std::atomic<int> refs{1};
    arkmpc_batch* parent = nullptr;      // a slice holds a reference on the batch that owns the storage
    int kind = 0, layout = 0, field_id = 0, device = 0;
    size_t n = 0;
    u32 words = 0;                       // u64 words per element in the arkworks AoS record
    u64* base = nullptr;                 // owned allocation (nullptr for slices)
    u64* share = nullptr;                // element i: share + stride * i   (the whole record for AoS kinds)
    u64* mac = nullptr;                  // ScalarShare batches only: MAC half of element i at mac + stride * i
    u32 stride = 0;                      // u64 units
    // asynchronous import (arkmpc_batch_from_host_async): the upload's completion, the pin on the caller's vector, a staging block
    hipEvent_t ready = nullptr;
    HostPins* pins = nullptr;
    void* staging = nullptr;
};

// ---- imports at link speed ------------------------------------------------------------------------------------------------------------
// n arkworks ScalarShare records where they lie (pinned host memory addressed over the link, or HBM) -> share column + MAC column, ONE
// pass: lane t moves quarter t of the records (a wave reads 1 KiB of contiguous source per access, fully coalesced 16-byte loads) and drops
// it at its place in the column.  Replaces upload-to-staging + split pass (64 B up, 64 B written, 128 B re-read / re-written per record).
// Few workgroups, as for the session kernels: the link, not the chip, is what the kernel waits for.
__global__ void __launch_bounds__(256) k_import_split(size_t n, const uint4* __restrict__ rec, uint4* __restrict__ share_col, uint4* __restrict__ mac_col) {
    const size_t total = 4 * n, step = (size_t)gridDim.x * 256;
    for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < total; q += step) {
        const uint4 v = rec[q];
        const size_t i = q >> 2;
        const unsigned k = (unsigned)(q & 3);
        (k < 2 ? share_col : mac_col)[2 * i + (k & 1)] = v;
    }
}
int ark_import_split(hipStream_t st, size_t n, const void* rec_dev, u64* share_col, u64* mac_col) {
    if (!n) return ARKMPC_OK;
    static const unsigned env = getenv("ARKMPC_IMPORT_BLOCKS") ? (unsigned)atoi(getenv("ARKMPC_IMPORT_BLOCKS")) : 0u;
    const size_t tiles = (4 * n + 255) / 256, want = env ? env : 128;
    hipLaunchKernelGGL(k_import_split, dim3((unsigned)(tiles < want ? tiles : want)), dim3(256), 0, st, n, (const uint4*)rec_dev, (uint4*)share_col, (uint4*)mac_col);
    return hipGetLastError() == hipSuccess ? ARKMPC_OK : ARKMPC_ERR_HIP;
}

static u32 elem_words(int kind, int field_id) {
    const bool ed = field_id == ARKMPC_CURVE25519_FR;
    switch (kind) {
        case ARKMPC_KIND_SCALAR: return 4;
        case ARKMPC_KIND_SCALAR_SHARE: return 8;
        case ARKMPC_KIND_POINT: return ed ? 16 : 12;
        case ARKMPC_KIND_POINT_SHARE: return ed ? 32 : 24;
        case ARKMPC_KIND_WORDS: return 1;
        default: return 0;
    }
}
static bool kind_ok_for_ctx(int kind, int field_id) {
    if (kind == ARKMPC_KIND_POINT || kind == ARKMPC_KIND_POINT_SHARE) return field_id == ARKMPC_BN254_FR || field_id == ARKMPC_CURVE25519_FR;
    return elem_words(kind, field_id) != 0;
}
static int batch_check(arkmpc_ctx* ctx, const arkmpc_batch* b) {
    if (!b) return ark_bad(ctx, "null batch");
    if (b->field_id != ctx->field_id || b->device != ctx->device) return ark_bad(ctx, "batch belongs to a different field / device");
    return ARKMPC_OK;
}
// The host side of an asynchronous import, given back: blocks until the upload has read the caller's records (the `ready` event needs no
// context), drops the pins, frees the staging block.  `o` is the OWNER of the storage; ctx is any context of its device.  The event returns to
// the context's free list only after a successful wait -- one that could not be waited for is destroyed, not recycled.
static int batch_release_host_side(arkmpc_ctx* ctx, arkmpc_batch* o) {
    int rc = ARKMPC_OK;
    if (o->ready) {
        if (hipEventSynchronize(o->ready) == hipSuccess) {
            CtxGuard guard(ctx);
            // consumers that did not call arkmpc_batch_acquire are still ordered from here on: the import is complete
            ctx->link_ev.push_back(o->ready);
        } else {
            (void)hipGetLastError();
            ark_set_err(ctx, "hipEventSynchronize(import) failed");
            rc = ARKMPC_ERR_HIP;
            (void)hipEventDestroy(o->ready);
        }
        o->ready = nullptr;
    }
    if (o->pins) { delete o->pins; o->pins = nullptr; }
    if (o->staging) { const int r = arkmpc_free(ctx, o->staging); if (r && !rc) rc = r; o->staging = nullptr; }
    return rc;
}

extern "C" {

int arkmpc_batch_create(arkmpc_ctx* ctx, int kind, int layout, size_t n, arkmpc_batch** out) {
    if (!ctx) return ARKMPC_ERR_BAD_ARG;
    if (!out) return ark_bad(ctx, "null out");
    *out = nullptr;
    if (ctx->host_buffers) return ark_bad(ctx, "batch handles live on device-pointer contexts");
    if (!kind_ok_for_ctx(kind, ctx->field_id)) return ark_bad(ctx, "element kind not defined for this context");
    if (layout != ARKMPC_LAYOUT_AOS && !(layout == ARKMPC_LAYOUT_SPLIT && kind == ARKMPC_KIND_SCALAR_SHARE))
        return ark_bad(ctx, "split layout is defined for ScalarShare batches only");
    const u32 w = elem_words(kind, ctx->field_id);
    if (n > (((size_t)1 << 48) / w)) return ark_bad(ctx, "batch too large");     // far above any HBM size; keeps n * w * 8 from wrapping
    void* p = nullptr;
    int rc = arkmpc_malloc(ctx, (n ? n : 1) * (size_t)w * 8 + 16, &p);
    if (rc) return rc;
    arkmpc_batch* b = new arkmpc_batch();
    b->kind = kind; b->layout = layout; b->field_id = ctx->field_id; b->device = ctx->device; b->n = n; b->words = w;
    b->base = (u64*)p;
    b->share = b->base;
    if (kind == ARKMPC_KIND_SCALAR_SHARE) {
        if (layout == ARKMPC_LAYOUT_SPLIT) { b->stride = 4; b->mac = b->base + 4 * n; }
        else { b->stride = 8; b->mac = b->base + 4; }
    } else {
        b->stride = w;
    }
    *out = b;
    return ARKMPC_OK;
}

int arkmpc_batch_retain(arkmpc_batch* b) {
    if (!b) return ARKMPC_ERR_BAD_ARG;
    b->refs.fetch_add(1, std::memory_order_relaxed);
    return ARKMPC_OK;
}

// Drops one reference.  The storage goes back to the pool -- stream-ordered on ctx's stream, like arkmpc_free -- when the
// last handle (the batch itself and every slice of it) is gone.
int arkmpc_batch_destroy(arkmpc_ctx* ctx, arkmpc_batch* b) {
    if (!ctx) return ARKMPC_ERR_BAD_ARG;
    if (b && b->device != ctx->device) return ark_bad(ctx, "batch lives on another device: destroy it through a context of its own device (nothing was released)");
    int rc = ARKMPC_OK;
    while (b && b->refs.fetch_sub(1, std::memory_order_acq_rel) == 1) {
        arkmpc_batch* parent = b->parent;
        // an import nobody released: wait for it, drop the pin, free the staging block -- unconditionally (round-5 advisor finding: the public
        // arkmpc_batch_host_release returns early for a context of another field, and the storage below must not go back to the pool while the
        // upload stream may still be writing it)
        if (b->ready || b->pins || b->staging) { int r = batch_release_host_side(ctx, b); if (r) rc = r; }
        if (b->base) { int r = arkmpc_free(ctx, b->base); if (r) rc = r; }
        delete b;
        b = parent;
    }
    return rc;
}

int arkmpc_batch_slice(arkmpc_ctx* ctx, arkmpc_batch* b, size_t lo, size_t count, arkmpc_batch** out) {
    if (!ctx) return ARKMPC_ERR_BAD_ARG;
    if (!out) return ark_bad(ctx, "null out");
    *out = nullptr;
    int rc = batch_check(ctx, b);
    if (rc) return rc;
    if (lo > b->n || count > b->n - lo) return ark_bad(ctx, "slice out of range");
    arkmpc_batch* s = new arkmpc_batch();
    s->kind = b->kind; s->layout = b->layout; s->field_id = b->field_id; s->device = b->device; s->n = count; s->words = b->words;
    s->stride = b->stride;
    s->share = b->share + (size_t)b->stride * lo;
    s->mac = b->mac ? b->mac + (size_t)b->stride * lo : nullptr;
    arkmpc_batch* owner = b->parent ? b->parent : b;         // slices of slices point at the owning batch
    owner->refs.fetch_add(1, std::memory_order_relaxed);
    s->parent = owner;
    *out = s;
    return ARKMPC_OK;
}

int arkmpc_batch_column(arkmpc_ctx* ctx, arkmpc_batch* b, int which, arkmpc_batch** out) {
    if (!ctx) return ARKMPC_ERR_BAD_ARG;
    if (!out) return ark_bad(ctx, "null out");
    *out = nullptr;
    int rc = batch_check(ctx, b);
    if (rc) return rc;
    if (b->kind != ARKMPC_KIND_SCALAR_SHARE || (which != 0 && which != 1)) return ark_bad(ctx, "column of a ScalarShare batch: which = 0 (share) or 1 (MAC)");
    if (b->layout != ARKMPC_LAYOUT_SPLIT) { ark_set_err(ctx, "columns of an AoS batch are strided: use arkmpc_share_extract"); return ARKMPC_ERR_UNSUPPORTED; }
    arkmpc_batch* s = new arkmpc_batch();
    s->kind = ARKMPC_KIND_SCALAR; s->layout = ARKMPC_LAYOUT_AOS; s->field_id = b->field_id; s->device = b->device; s->n = b->n; s->words = 4;
    s->stride = 4;
    s->share = which ? b->mac : b->share;
    arkmpc_batch* owner = b->parent ? b->parent : b;
    owner->refs.fetch_add(1, std::memory_order_relaxed);
    s->parent = owner;
    *out = s;
    return ARKMPC_OK;
}

size_t arkmpc_batch_len(const arkmpc_batch* b) { return b ? b->n : 0; }
int arkmpc_batch_kind(const arkmpc_batch* b) { return b ? b->kind : -1; }
int arkmpc_batch_layout(const arkmpc_batch* b) { return b ? b->layout : -1; }
size_t arkmpc_batch_elem_words(const arkmpc_batch* b) { return b ? b->words : 0; }
uint64_t* arkmpc_batch_data(const arkmpc_batch* b) { return b ? b->share : nullptr; }
uint64_t* arkmpc_batch_mac_data(const arkmpc_batch* b) { return b ? b->mac : nullptr; }
size_t arkmpc_batch_stride(const arkmpc_batch* b) { return b ? b->stride : 0; }

// host Vec<T> (arkworks records, n * elem_words u64) -> a new device batch in the requested layout
int arkmpc_batch_from_host(arkmpc_ctx* ctx, int kind, int layout, size_t n, const void* host_records, arkmpc_batch** out) {
    if (!ctx) return ARKMPC_ERR_BAD_ARG;
    if (n && !host_records) return ark_bad(ctx, "null host records");
    int rc = arkmpc_batch_create(ctx, kind, layout, n, out);
    if (rc) return rc;
    arkmpc_batch* b = *out;
    const size_t bytes = n * (size_t)b->words * 8;
    if (!n) return ARKMPC_OK;
    if (layout == ARKMPC_LAYOUT_AOS) {
        rc = arkmpc_memcpy_h2d(ctx, b->base, host_records, bytes);
    } else {                                                   // upload the records, split the columns on the device
        void* tmp = nullptr;
        rc = arkmpc_malloc(ctx, bytes, &tmp);
        if (!rc) rc = arkmpc_memcpy_h2d(ctx, tmp, host_records, bytes);
        if (!rc) rc = arkmpc_share_split(ctx, n, (const u64*)tmp, b->share, b->mac);
        if (tmp) { int r = arkmpc_free(ctx, tmp); if (!rc) rc = r; }
    }
    if (rc) { arkmpc_batch_destroy(ctx, b); *out = nullptr; }
    return rc;
}

// The same without blocking: the records go up on the context's `up` stream -- read IN PLACE over the link by k_import_split when the batch
// is split columns, plain DMA otherwise -- behind whatever the compute stream is doing (the previous gate's kernels, the network round).
// The caller's vector is pinned in place if it is not pinned already.  Small or unpinnable vectors take the blocking path above.
int arkmpc_batch_from_host_async(arkmpc_ctx* ctx, int kind, int layout, size_t n, const void* host_records, arkmpc_batch** out) {
    if (!ctx) return ARKMPC_ERR_BAD_ARG;
    if (n && !host_records) return ark_bad(ctx, "null host records");
    int rc = arkmpc_batch_create(ctx, kind, layout, n, out);
    if (rc) return rc;
    arkmpc_batch* b = *out;
    if (!n) return ARKMPC_OK;
    const size_t bytes = n * (size_t)b->words * 8;
    const bool split = layout == ARKMPC_LAYOUT_SPLIT;
    HostPins* pins = new HostPins();
    Place src = classify_and_hold(ctx, *pins, host_records, bytes);
    if (src.kind == Mem::Pageable) { pins->pin(host_records, bytes); src = classify(ctx, host_records, bytes); }
    if (src.kind == Mem::Foreign || (src.kind == Mem::Device && ((uintptr_t)host_records & 15))) {
        delete pins; arkmpc_batch_destroy(ctx, b); *out = nullptr;
        return ark_bad(ctx, "records in device memory must be 16-byte aligned and on the context's GPU");
    }
    void* staging = nullptr;
    if (src.kind != Mem::Pageable && split && !src.zc()) rc = arkmpc_malloc(ctx, bytes, &staging);      // pinned but only 8-byte aligned: DMA, then the columns from HBM
    if (src.kind == Mem::Pageable || rc) {               // not pinnable (below ARKMPC_PIN_MIN_KB, a read-only mapping ...): the blocking import
        delete pins;
        arkmpc_batch_destroy(ctx, b); *out = nullptr;
        if (rc) return rc;
        rc = arkmpc_batch_from_host(ctx, kind, layout, n, host_records, out);
        if (!rc) { CtxGuard g(ctx); ctx->stats.batch_blocking_imports++; }
        return rc;
    }
    {
        CtxGuard guard(ctx);
        rc = guard.rc;
        if (!rc) rc = link_ensure(ctx);
        LinkEvents evs(ctx);
        if (!rc) rc = evs.edge(ctx->stream, ctx->up);        // the batch's block (and the staging block) were last used on the compute stream
        if (!rc) {
            hipError_t e = hipSuccess;
            if (split && src.zc()) rc = ark_import_split(ctx->up, n, src.dev, b->share, b->mac);
            else if (split) {
                e = hipMemcpyAsync(staging, host_records, bytes, hipMemcpyDefault, ctx->up);
                if (e == hipSuccess) rc = ark_import_split(ctx->up, n, staging, b->share, b->mac);
            } else e = hipMemcpyAsync(b->base, host_records, bytes, hipMemcpyDefault, ctx->up);
            if (e != hipSuccess) { ark_set_err(ctx, std::string("asynchronous import: ") + hipGetErrorString(e)); rc = ARKMPC_ERR_HIP; }
            else if (rc) ark_set_err(ctx, "import kernel launch failed");
        }
        hipEvent_t ready = nullptr;
        if (!rc) rc = evs.mark(ctx->up, &ready);
        if (!rc) {
            evs.used.pop_back();                             // `ready` stays with the batch; the edge's event goes back to the free list now (its wait is enqueued)
            b->ready = ready; b->pins = pins; b->staging = staging;
            ctx->stats.batch_async_imports++;
        }
        evs.give_back();
    }
    if (rc) {
        (void)hipStreamSynchronize(ctx->up);
        delete pins;
        if (staging) (void)arkmpc_free(ctx, staging);
        arkmpc_batch_destroy(ctx, b); *out = nullptr;
    }
    return rc;
}
static arkmpc_batch* owner_of(arkmpc_batch* b) { return b->parent ? b->parent : b; }
// the context's compute stream waits -- on the device -- for the batch's import; returns at once.  Call it before the first use of the batch
// on this context (the batch-level entry points below do it themselves).  A batch that was not imported asynchronously: nothing to do.
int arkmpc_batch_acquire(arkmpc_ctx* ctx, arkmpc_batch* batch) {
    if (!ctx) return ARKMPC_ERR_BAD_ARG;
    int rc = batch_check(ctx, batch);
    if (rc) return rc;
    arkmpc_batch* o = owner_of(batch);
    if (!o->ready) return ARKMPC_OK;
    CtxGuard guard(ctx);
    if (guard.rc) return guard.rc;
    ARK_HIP(ctx, hipStreamWaitEvent(ctx->stream, o->ready, 0));
    return ARKMPC_OK;
}
// blocks until the import has read the caller's vector to its end, then drops the pin: the vector may be freed or overwritten.  (Consumers
// enqueued after arkmpc_batch_acquire stay ordered by the stream.)  Idempotent.
int arkmpc_batch_host_release(arkmpc_ctx* ctx, arkmpc_batch* batch) {
    if (!ctx) return ARKMPC_ERR_BAD_ARG;
    int rc = batch_check(ctx, batch);
    if (rc) return rc;
    return batch_release_host_side(ctx, owner_of(batch));
}

// the batch as the arkworks Vec<T> the awaiting caller expects (always AoS records); blocks until the data has landed
int arkmpc_batch_to_host(arkmpc_ctx* ctx, const arkmpc_batch* b, void* host_records_out) {
    if (!ctx) return ARKMPC_ERR_BAD_ARG;
    int rc = batch_check(ctx, b);
    if (rc) return rc;
    if ((rc = arkmpc_batch_acquire(ctx, const_cast<arkmpc_batch*>(b)))) return rc;
    if (b->n && !host_records_out) return ark_bad(ctx, "null host buffer");
    if (!b->n) return arkmpc_sync(ctx);
    const size_t bytes = b->n * (size_t)b->words * 8;
    if (b->layout == ARKMPC_LAYOUT_AOS) return arkmpc_memcpy_d2h(ctx, host_records_out, b->share, bytes);
    void* tmp = nullptr;
    rc = arkmpc_malloc(ctx, bytes, &tmp);
    if (!rc) rc = arkmpc_share_join(ctx, b->n, b->share, b->mac, (u64*)tmp);
    if (!rc) rc = arkmpc_memcpy_d2h(ctx, host_records_out, tmp, bytes);
    if (tmp) { int r = arkmpc_free(ctx, tmp); if (!rc) rc = r; }
    return rc;
}

// ---- batch-level forms of the Beaver multiplication (authenticated_scalar.rs:848-879): operands and results are handles, the
// ---- layout tag selects the AoS or the split-column kernels, nothing leaves HBM --------------------------------------------
static int share_batch_check(arkmpc_ctx* ctx, const arkmpc_batch* b, size_t n) {
    int rc = batch_check(ctx, b);
    if (rc) return rc;
    if (b->kind != ARKMPC_KIND_SCALAR_SHARE) return ark_bad(ctx, "expected a ScalarShare batch");
    if (b->n != n) return ark_bad(ctx, "Cannot operate on batches of different sizes");
    return ARKMPC_OK;
}
// K1: out_de = a new Scalar batch of 2n elements, d then e -- the payload this party sends (:863-868, :141-145)
int arkmpc_batch_beaver_mask(arkmpc_ctx* ctx, const arkmpc_batch* x, const arkmpc_batch* y, const arkmpc_batch* a, const arkmpc_batch* b,
                             arkmpc_batch** out_de) {
    if (!ctx) return ARKMPC_ERR_BAD_ARG;
    if (!out_de) return ark_bad(ctx, "null out");
    *out_de = nullptr;
    if (!x) return ark_bad(ctx, "null batch");
    const size_t n = x->n;
    int rc;
    if ((rc = share_batch_check(ctx, x, n)) || (rc = share_batch_check(ctx, y, n)) || (rc = share_batch_check(ctx, a, n)) || (rc = share_batch_check(ctx, b, n))) return rc;
    for (const arkmpc_batch* q : {x, y, a, b}) if ((rc = arkmpc_batch_acquire(ctx, const_cast<arkmpc_batch*>(q)))) return rc;
    if ((rc = arkmpc_batch_create(ctx, ARKMPC_KIND_SCALAR, ARKMPC_LAYOUT_AOS, 2 * n, out_de))) return rc;
    rc = arkmpc_beaver_mask_v(ctx, n, x->share, x->stride, y->share, y->stride, a->share, a->stride, b->share, b->stride, (*out_de)->share);
    if (rc) { arkmpc_batch_destroy(ctx, *out_de); *out_de = nullptr; }
    return rc;
}
// K2+K3: result = a new ScalarShare batch in `out_layout`
int arkmpc_batch_beaver_finish(arkmpc_ctx* ctx, int party_id, const uint64_t mac_key[4], const arkmpc_batch* my_de, const arkmpc_batch* peer_de,
                               const arkmpc_batch* a, const arkmpc_batch* b, const arkmpc_batch* c, int out_layout, arkmpc_batch** out) {
    if (!ctx) return ARKMPC_ERR_BAD_ARG;
    if (!out) return ark_bad(ctx, "null out");
    *out = nullptr;
    if (!a) return ark_bad(ctx, "null batch");
    const size_t n = a->n;
    int rc;
    if ((rc = share_batch_check(ctx, a, n)) || (rc = share_batch_check(ctx, b, n)) || (rc = share_batch_check(ctx, c, n))) return rc;
    if ((rc = batch_check(ctx, my_de)) || (rc = batch_check(ctx, peer_de))) return rc;
    if (my_de->kind != ARKMPC_KIND_SCALAR || peer_de->kind != ARKMPC_KIND_SCALAR || my_de->n != 2 * n || peer_de->n != 2 * n)
        return ark_bad(ctx, "d||e batches must hold 2n Scalars");          // a short peer payload is an error, never an out-of-bounds read
    for (const arkmpc_batch* q : {my_de, peer_de, a, b, c}) if ((rc = arkmpc_batch_acquire(ctx, const_cast<arkmpc_batch*>(q)))) return rc;
    if ((rc = arkmpc_batch_create(ctx, ARKMPC_KIND_SCALAR_SHARE, out_layout, n, out))) return rc;
    arkmpc_batch* r = *out;
    rc = arkmpc_beaver_finish_fused_v(ctx, n, party_id, mac_key, my_de->share, peer_de->share, a->share, a->mac, a->stride, b->share, b->mac, b->stride,
                                      c->share, c->mac, c->stride, r->share, r->mac, r->stride);
    if (rc) { arkmpc_batch_destroy(ctx, r); *out = nullptr; }
    return rc;
}

// ---- composite gate: the point-side K3 (authenticated_curve.rs:703-713), regrouped -- see include/arkmpc.h --------------------------------
}  // extern "C"
namespace {
// scratch in the address space the context's buffers live in (device memory, or host memory for a host-buffer context)
struct Tmp {
    arkmpc_ctx* ctx; void* p = nullptr; int rc = ARKMPC_OK;
    Tmp(arkmpc_ctx* c, size_t bytes) : ctx(c) {
        if (c->host_buffers) { p = std::malloc(bytes ? bytes : 16); if (!p) { ark_set_err(c, "out of host memory for a temporary of the composite gate"); rc = ARKMPC_ERR_BAD_ARG; } }
        else rc = arkmpc_malloc(c, bytes ? bytes : 16, &p);
    }
    ~Tmp() { if (p) { if (ctx->host_buffers) std::free(p); else arkmpc_free(ctx, p); } }
    u64* u() const { return (u64*)p; }
};
typedef int (*MulPointFn)(arkmpc_ctx*, size_t, const uint64_t*, const uint64_t*, uint64_t*);
typedef int (*MulGenFn)(arkmpc_ctx*, size_t, const uint64_t*, uint64_t*);
typedef int (*AddFn)(arkmpc_ctx*, size_t, const uint64_t*, const uint64_t*, uint64_t*);
int point_beaver_finish(arkmpc_ctx* ctx, size_t n, int party, const uint64_t key[4], const uint64_t* d, const uint64_t* eG, const uint64_t* a,
                        const uint64_t* b, const uint64_t* c, uint64_t* out, size_t pw, MulPointFn mul_point, MulGenFn mul_gen, AddFn add) {
    if (!ctx) return ARKMPC_ERR_BAD_ARG;
    if (party != 0 && party != 1) return ark_bad(ctx, "party_id must be 0 or 1");
    if (!key) return ark_bad(ctx, "null mac_key");
    if (n && (!d || !eG || !a || !b || !c || !out)) return ark_bad(ctx, "null buffer");
    if (!n) return ARKMPC_OK;
    if (n > (((size_t)1 << 48) / (2 * pw))) return ark_bad(ctx, "batch too large");           // the guard of arkmpc_batch_create: n * 2 * pw * 8 cannot wrap
    Tmp on_eG(ctx, n * 64), on_G(ctx, n * 64), t1(ctx, n * 2 * pw * 8);
    if (on_eG.rc || on_G.rc || t1.rc) return on_eG.rc ? on_eG.rc : (on_G.rc ? on_G.rc : t1.rc);
    int rc = arkmpc_share_add_public(ctx, n, party, key, a, d, on_eG.u());                     // [a] + d
    if (!rc) rc = arkmpc_share_mul_public(ctx, n, b, d, on_G.u());                             // d [b]
    if (!rc) rc = arkmpc_share_add(ctx, n, on_G.u(), c, on_G.u());                             // [c] + d [b]
    if (!rc) rc = mul_point(ctx, n, on_eG.u(), eG, t1.u());                                    // ([a] + d) * eG
    if (!rc) rc = mul_gen(ctx, n, on_G.u(), out);                                              // ([c] + d [b]) * G
    if (!rc) rc = add(ctx, n, t1.u(), out, out);
    return rc;
}
}  // namespace
extern "C" {
int arkmpc_point_beaver_finish(arkmpc_ctx* ctx, size_t n, int party_id, const uint64_t mac_key[4], const uint64_t* d_open, const uint64_t* eG_open,
                               const uint64_t* a, const uint64_t* b, const uint64_t* c, uint64_t* out_shares) {
    return point_beaver_finish(ctx, n, party_id, mac_key, d_open, eG_open, a, b, c, out_shares, 12, arkmpc_scalarshare_mul_point,
                               arkmpc_scalarshare_mul_generator, arkmpc_pointshare_add);
}
int arkmpc_edpoint_beaver_finish(arkmpc_ctx* ctx, size_t n, int party_id, const uint64_t mac_key[4], const uint64_t* d_open, const uint64_t* eG_open,
                                 const uint64_t* a, const uint64_t* b, const uint64_t* c, uint64_t* out_shares) {
    return point_beaver_finish(ctx, n, party_id, mac_key, d_open, eG_open, a, b, c, out_shares, 16, arkmpc_scalarshare_mul_ed_point,
                               arkmpc_scalarshare_mul_ed_generator, arkmpc_edshare_add);
}

// ---- events: device-side ordering between the streams of two contexts ---------------------------------------------------------
}  // extern "C"
struct arkmpc_event { hipEvent_t ev; int device; };
// hipEventCreate / hipEventDestroy cost microseconds each and a latency-bound circuit records one event per message: retired events are
// kept per device and recorded again.  Re-recording is safe once the consumers have ISSUED their waits (hipStreamWaitEvent binds to the
// record that is current when it is called), which is what arkmpc_event_destroy's contract already requires.
namespace {
std::mutex g_ev_mu;
std::vector<arkmpc_event*> g_ev_free[16];
}
extern "C" {
int arkmpc_event_record(arkmpc_ctx* ctx, arkmpc_event** out) {
    if (!ctx) return ARKMPC_ERR_BAD_ARG;
    CtxGuard guard(ctx);
    if (guard.rc) return guard.rc;
    if (!out) return ark_bad(ctx, "null out");
    *out = nullptr;
    arkmpc_event* e = nullptr;
    if (ctx->device < 16) {
        std::lock_guard<std::mutex> lk(g_ev_mu);
        auto& fl = g_ev_free[ctx->device];
        if (!fl.empty()) { e = fl.back(); fl.pop_back(); }
    }
    if (!e) {
        hipEvent_t ev;
        ARK_HIP(ctx, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        e = new arkmpc_event{ev, ctx->device};
    }
    hipError_t err = hipEventRecord(e->ev, ctx->stream);
    if (err != hipSuccess) { (void)hipEventDestroy(e->ev); delete e; ark_set_err(ctx, std::string("hipEventRecord: ") + hipGetErrorString(err)); return ARKMPC_ERR_HIP; }
    *out = e;
    return ARKMPC_OK;
}
int arkmpc_event_wait(arkmpc_ctx* ctx, arkmpc_event* event) {
    if (!ctx) return ARKMPC_ERR_BAD_ARG;
    CtxGuard guard(ctx);
    if (guard.rc) return guard.rc;
    if (!event) return ark_bad(ctx, "null event");
    ARK_HIP(ctx, hipStreamWaitEvent(ctx->stream, event->ev, 0));
    return ARKMPC_OK;
}
int arkmpc_event_destroy(arkmpc_event* event) {
    if (!event) return ARKMPC_ERR_BAD_ARG;
    if (event->device >= 0 && event->device < 16) {
        std::lock_guard<std::mutex> lk(g_ev_mu);
        if (g_ev_free[event->device].size() < 4096) { g_ev_free[event->device].push_back(event); return ARKMPC_OK; }
    }
    (void)hipEventDestroy(event->ev);          // HIP defers the release until the event has completed
    delete event;
    return ARKMPC_OK;
}

}  // extern "C"
