/*
 * arkmpc.h -- C ABI of the MI355X-native engine for ark-mpc's batched authenticated-share path.
 *
 * This is the drop-in boundary: the entry points a patched ark-mpc `batch_*` gate closure
 * (online-phase/src/fabric.rs:841-854 `new_batch_gate_op`) would bind over Rust FFI -- see
 * INTEGRATION.md for the `extern "C"` shim.  Plain pointers and sizes only; no C++/torch types.
 *
 * DATA LAYOUT (identical to arkworks' in-memory layout, so a Rust `&[T]` can be passed as-is):
 *   Scalar<C>       4 x u64  little-endian limbs, Montgomery form, R = 2^256   (scalar.rs:46)
 *   ScalarShare<C>  8 x u64  = { share[4], mac[4] }                             (scalar/share.rs:32-37)
 *   CurvePoint<C>   12 x u64 = short-Weierstrass Jacobian { x[4], y[4], z[4] }, identity z = 0
 *                   (ark-ec `Projective`; BN254 G1 only)                        (curve/curve.rs:47)
 *   PointShare<C>   24 x u64 = { share[12], mac[12] }                           (curve/share.rs:25-30)
 * "share view" entry points (suffix _v) take the share and MAC columns as separate base pointers
 * with an element stride, so the engine-native split layout (all shares, then all MACs; stride 4)
 * and the arkworks AoS layout (stride 8, mac = share + 4) run through the same kernels.
 *
 * MEMORY SPACE: by default every buffer pointer is a DEVICE pointer (16-byte aligned) on the
 * context's GPU and calls are asynchronous on the context's stream.  After
 * arkmpc_ctx_set_host_buffers(ctx, 1) every buffer pointer is a HOST pointer; the call stages
 * through device scratch and returns when the outputs are in host memory.
 * `mac_key`, `blinder` and single-element outputs are always host pointers to 4 x u64.
 *
 * ERRORS: every function returns ARKMPC_OK (0) or a negative arkmpc_status; nothing throws or
 * aborts across the ABI (the reference's closures are infallible and panic on misuse,
 * fabric/result.rs:127-233; here misuse is a status code).  A MAC-check failure is a VALUE
 * (out_ok = 0), surfaced by the caller as MpcError::AuthenticationError
 * (authenticated_scalar.rs:368-385), not an error status.
 *
 * THREADING: a context may be used from any thread, one call at a time per context (calls on one
 * context are serialised by an internal mutex); different contexts are independent -- this matches
 * gates running concurrently on rayon workers (fabric/executor/multi_threaded/executor.rs:208-217).
 */
#ifndef ARKMPC_H
#define ARKMPC_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct arkmpc_ctx arkmpc_ctx;

typedef enum {
    ARKMPC_OK = 0,
    ARKMPC_ERR_BAD_ARG = -1,      /* null pointer, bad field/party id, misaligned device pointer */
    ARKMPC_ERR_HIP = -2,          /* HIP runtime failure; see arkmpc_last_error */
    ARKMPC_ERR_UNSUPPORTED = -3,  /* op not defined for this context's field (e.g. curve ops on Fr ctx) */
    ARKMPC_ERR_NO_DEVICE = -4     /* no usable GPU: the engine has no CPU fallback */
} arkmpc_status;

typedef enum {
    ARKMPC_BN254_FR = 0,       /* scalar field of BN254 (ark-bn254, the reference's TestCurve: lib.rs:78) */
    ARKMPC_BLS12_381_FR = 1,   /* scalar field of BLS12-381 */
    ARKMPC_CURVE25519_FR = 2,  /* ed25519 group order (README.md:24 uses ark-curve25519) */
    ARKMPC_BN254_FQ = 3,       /* base field of BN254 (coordinates of G1 points) */
    ARKMPC_CURVE25519_FQ = 4   /* 2^255 - 19 (coordinates of Curve25519 / ed25519 points) */
} arkmpc_field;

/* ---- context ------------------------------------------------------------------------------ */
/* One context = one (field, GPU) pair with its own stream and scratch.  For point ops create the
 * context with ARKMPC_BN254_FR: scalars are Fr elements, coordinates are Fq elements. */
int arkmpc_ctx_create(int field_id, int device, arkmpc_ctx** out_ctx);
int arkmpc_ctx_destroy(arkmpc_ctx* ctx);
/* Use a caller-owned hipStream_t (e.g. torch's current stream) instead of the context's own. */
int arkmpc_ctx_set_stream(arkmpc_ctx* ctx, void* hip_stream);
int arkmpc_ctx_set_host_buffers(arkmpc_ctx* ctx, int enabled);
int arkmpc_sync(arkmpc_ctx* ctx);
const char* arkmpc_last_error(arkmpc_ctx* ctx);
/* Counters of one context since its creation (diagnostics: which path the streaming sessions below took, what they held).  A session phase
 * counts once, under the path that ran it; *_bytes_* = device memory one session held (the last one to take a block / the largest so far). */
typedef struct {
    uint64_t hostmul_zero_copy_phases[2];   /* [0] phase 1 (_begin), [1] phase 2 (_finish): ran as ONE kernel on the caller's vectors in place */
    uint64_t hostmul_copy_phases[2];        /* ... ran through the three-stream copy pipeline */
    uint64_t hostmul_device_bytes_last;
    uint64_t hostmul_device_bytes_peak;
    uint64_t batch_async_imports;           /* arkmpc_batch_from_host_async calls that went up asynchronously (in-place kernel or DMA) */
    uint64_t batch_blocking_imports;        /* ... that fell back to the blocking copy (small or unpinnable vectors) */
    uint64_t zc_refused_reused_address;     /* caller-pinned vectors a kernel could have addressed in place but did NOT, because the vector was registered
                                             * (arkmpc_host_register) at addresses that already had a registered life which ended: they travelled by DMA */
} arkmpc_ctx_stats;
int arkmpc_ctx_get_stats(arkmpc_ctx* ctx, arkmpc_ctx_stats* out_stats);
/* Kernel timer: arm slot s (0..63) and the NEXT K1 / K2+K3 / K5 launch on this context gets HIP events bound to
 * its dispatch (hipExtLaunchKernelGGL start/stop events): arkmpc_kernel_timer_ms then returns that kernel's own duration
 * (it blocks until the kernel has finished).  Unlike marker events recorded between launches this excludes the dispatch gap. */
int arkmpc_kernel_timer_arm(arkmpc_ctx* ctx, int slot);
int arkmpc_kernel_timer_ms(arkmpc_ctx* ctx, int slot, float* out_ms);
/* Cross-context ordering without blocking the host: arkmpc_event_record marks the work submitted so far on ctx's stream,
 * arkmpc_event_wait makes ANOTHER context's stream wait for that mark on the device (the host call returns at once).  This is
 * how a batch handed from one party / gate to another stays ordered when both keep their own stream (the mock link of the
 * host mirror; rayon workers sharing device batches).  The event may be waited on by any number of contexts, then destroyed. */
typedef struct arkmpc_event arkmpc_event;
int arkmpc_event_record(arkmpc_ctx* ctx, arkmpc_event** out_event);
int arkmpc_event_wait(arkmpc_ctx* ctx, arkmpc_event* event);
int arkmpc_event_destroy(arkmpc_event* event);
const char* arkmpc_version(void);
int arkmpc_device_count(void);
/* device memory helpers for callers without their own allocator (Rust shim, tests).  Blocks come from a per-device pool;
 * arkmpc_free is STREAM-ORDERED like hipFreeAsync: the block is recycled once the work already submitted on ctx's stream
 * has passed, without blocking the host -- free a buffer through the context whose stream used it last. */
int arkmpc_malloc(arkmpc_ctx* ctx, size_t bytes, void** out_dptr);
int arkmpc_free(arkmpc_ctx* ctx, void* dptr);
int arkmpc_memcpy_h2d(arkmpc_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes);
int arkmpc_memcpy_d2h(arkmpc_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes);
int arkmpc_memcpy_d2d(arkmpc_ctx* ctx, void* dst_dev, const void* src_dev, size_t bytes);   /* asynchronous on the context's stream */

/* ---- device-batch carrier: what a `ResultValue::DeviceBatch` variant holds (fabric/result.rs:47-64) -------------------------
 * An opaque, reference-counted handle to ONE batch value resident in HBM: element kind (the ResultValue variant it stands
 * for), element count, layout tag and storage.  Gate outputs stay on the GPU between gates as handles; the arkworks Vec<T> is
 * materialised only where a caller awaits the value (arkmpc_batch_to_host).  Handles belong to (field, device) of the creating
 * context; any context of that pair may use them.  Batch contexts take device pointers (not arkmpc_ctx_set_host_buffers). */
typedef struct arkmpc_batch arkmpc_batch;
typedef enum {
    ARKMPC_KIND_SCALAR = 0,        /* ResultValue::ScalarBatch  -- Vec<Scalar<C>>,      4 x u64 per element            */
    ARKMPC_KIND_SCALAR_SHARE = 1,  /* Vec of ResultValue::ScalarShare -- ScalarShare<C>, 8 x u64                         */
    ARKMPC_KIND_POINT = 2,         /* ResultValue::PointBatch   -- Vec<CurvePoint<C>>, 12 (BN254 G1) / 16 (Curve25519)  */
    ARKMPC_KIND_POINT_SHARE = 3,   /* Vec of ResultValue::PointShare -- PointShare<C>,  24 / 32 x u64                    */
    ARKMPC_KIND_WORDS = 4          /* ResultValue::Bytes and engine scratch: n raw u64 words                            */
} arkmpc_kind;
typedef enum {
    ARKMPC_LAYOUT_AOS = 0,         /* arkworks records as a Rust Vec<T> holds them                                      */
    ARKMPC_LAYOUT_SPLIT = 1        /* ScalarShare batches only: n shares, then n MACs (the engine-native columns)       */
} arkmpc_layout;
int arkmpc_batch_create(arkmpc_ctx* ctx, int kind, int layout, size_t n, arkmpc_batch** out_batch);   /* uninitialised storage */
/* upload a host Vec<T> (n arkworks records); with ARKMPC_LAYOUT_SPLIT the columns are separated on the device */
int arkmpc_batch_from_host(arkmpc_ctx* ctx, int kind, int layout, size_t n, const void* host_records, arkmpc_batch** out_batch);
/* The same WITHOUT blocking, at link speed -- for the triples a PreprocessingPhase hands over as host vectors gate after gate
 * (fabric.rs:894-915 next_triple_batch, offline_prep.rs:65-81; 192 B per party-gate, what bounds a circuit whose operands are resident):
 * the records go up on the context's upload stream behind whatever its compute stream is doing.  ScalarShare batches in
 * ARKMPC_LAYOUT_SPLIT are read IN PLACE over the link by one kernel that writes the two columns (no staging copy, no split pass); everything
 * else is one DMA.  The in-place kernel is for vectors the caller holds in pinned memory (arkmpc_host_alloc / arkmpc_host_register); a pageable
 * vector is registered for the call and goes up by DMA into a staging block that the split kernel then reads (no kernel addresses a vector the
 * library registered itself, see the streaming sessions).
 *   _acquire        the context's compute stream waits, on the device, for the import (returns at once).  The batch-level entry points do
 *                   it themselves; call it before handing arkmpc_batch_data() pointers to the pointer-level ones.
 *   _host_release   blocks until the import has read host_records to its end and drops the pin; only then may the vector be freed or
 *                   overwritten (arkmpc_batch_destroy does the same if it was never called).
 * Vectors that cannot be pinned (below 1 MiB, read-only mappings) take the blocking path of arkmpc_batch_from_host.
 * A handle's import state belongs to the thread that drives the batch: _acquire / _host_release / _destroy of ONE handle are not to be
 * called concurrently (different handles, and retained clones once the import has been released, are independent). */
int arkmpc_batch_from_host_async(arkmpc_ctx* ctx, int kind, int layout, size_t n, const void* host_records, arkmpc_batch** out_batch);
int arkmpc_batch_acquire(arkmpc_ctx* ctx, arkmpc_batch* batch);
int arkmpc_batch_host_release(arkmpc_ctx* ctx, arkmpc_batch* batch);
/* the batch as n arkworks records in host memory, whatever its device layout; blocks */
int arkmpc_batch_to_host(arkmpc_ctx* ctx, const arkmpc_batch* batch, void* host_records_out);
/* &v[lo .. lo+count] as a new handle that SHARES the storage (and keeps it alive): index-range sharding, sub-batches */
int arkmpc_batch_slice(arkmpc_ctx* ctx, arkmpc_batch* batch, size_t lo, size_t count, arkmpc_batch** out_batch);
int arkmpc_batch_retain(arkmpc_batch* batch);                      /* Clone of the handle: one more reference */
/* Drop of one handle; the storage returns to the pool (ordered on ctx's stream, like arkmpc_free) with the last reference */
int arkmpc_batch_destroy(arkmpc_ctx* ctx, arkmpc_batch* batch);
/* the share (which = 0) or MAC (1) column of a ScalarShare batch in ARKMPC_LAYOUT_SPLIT as a Scalar batch that SHARES the storage:
 * the `.share()` projection open_batch sends (authenticated_scalar.rs:141-145) without a copy.  AoS batches: ARKMPC_ERR_UNSUPPORTED
 * (their columns are strided; use arkmpc_share_extract). */
int arkmpc_batch_column(arkmpc_ctx* ctx, arkmpc_batch* batch, int which, arkmpc_batch** out_batch);
size_t arkmpc_batch_len(const arkmpc_batch* batch);
int arkmpc_batch_kind(const arkmpc_batch* batch);
int arkmpc_batch_layout(const arkmpc_batch* batch);
size_t arkmpc_batch_elem_words(const arkmpc_batch* batch);         /* u64 words of one arkworks record */
/* device addresses for the pointer-level entry points below: element i starts at data + stride * i (u64 units); for
 * ScalarShare batches `data` addresses the share half and `mac_data` the MAC half (NULL for other kinds) */
uint64_t* arkmpc_batch_data(const arkmpc_batch* batch);
uint64_t* arkmpc_batch_mac_data(const arkmpc_batch* batch);
size_t arkmpc_batch_stride(const arkmpc_batch* batch);
/* Beaver multiplication on handles (authenticated_scalar.rs:848-879): K1 returns this party's d||e payload as a new Scalar
 * batch of 2n elements; K2+K3 returns the product shares as a new ScalarShare batch in `out_layout`.  Operand layouts may be
 * mixed; a d||e batch of the wrong length is ARKMPC_ERR_BAD_ARG. */
int arkmpc_batch_beaver_mask(arkmpc_ctx* ctx, const arkmpc_batch* x, const arkmpc_batch* y, const arkmpc_batch* a,
                             const arkmpc_batch* b, arkmpc_batch** out_de);
int arkmpc_batch_beaver_finish(arkmpc_ctx* ctx, int party_id, const uint64_t mac_key[4], const arkmpc_batch* my_de,
                               const arkmpc_batch* peer_de, const arkmpc_batch* a, const arkmpc_batch* b, const arkmpc_batch* c,
                               int out_layout, arkmpc_batch** out_batch);

/* ---- Scalar<C> vectors: scalar.rs:210-267, scalar_result.rs:24-278 ------------------------- */
int arkmpc_scalar_add(arkmpc_ctx* ctx, size_t n, const uint64_t* a, const uint64_t* b, uint64_t* out);
int arkmpc_scalar_sub(arkmpc_ctx* ctx, size_t n, const uint64_t* a, const uint64_t* b, uint64_t* out);
int arkmpc_scalar_mul(arkmpc_ctx* ctx, size_t n, const uint64_t* a, const uint64_t* b, uint64_t* out); /* ScalarResult::batch_mul, scalar_result.rs:257-278 */
int arkmpc_scalar_neg(arkmpc_ctx* ctx, size_t n, const uint64_t* a, uint64_t* out);
/* Scalar::batch_inverse / ScalarResult::batch_inverse (scalar.rs:93-100, scalar_result.rs:50-61): non-zero elements are
 * replaced by their inverses, zeros stay zero (ark_ff::batch_inversion).  In-place allowed. */
int arkmpc_scalar_batch_inverse(arkmpc_ctx* ctx, size_t n, const uint64_t* a, uint64_t* out);
/* inclusive prefix products out_i = a_0 * ... * a_i: the public scan inside the prefix_product gadget (gadgets.rs:131-137),
 * there a sequential chain of ScalarResult multiplications; here a parallel scan (multiplication is associative). */
int arkmpc_scalar_prefix_product(arkmpc_ctx* ctx, size_t n, const uint64_t* a, uint64_t* out);
/* reductions to ONE element (`out` follows the context's buffer mode like every other buffer): Sum / Product for ScalarResult
 * (scalar_result.rs:325-338 and Iterator::sum); n = 0 gives 0 / 1.  The reference folds left to right; + and * are associative and
 * commutative on canonical residues, so the tree used here returns the same words. */
int arkmpc_scalar_sum(arkmpc_ctx* ctx, size_t n, const uint64_t* a, uint64_t* out);
int arkmpc_scalar_product(arkmpc_ctx* ctx, size_t n, const uint64_t* a, uint64_t* out);
/* canonical little-endian integers (< 2^256, reduced mod p on the way in) <-> Montgomery form */
int arkmpc_scalar_from_canonical(arkmpc_ctx* ctx, size_t n, const uint64_t* in, uint64_t* out);
int arkmpc_scalar_to_canonical(arkmpc_ctx* ctx, size_t n, const uint64_t* in, uint64_t* out);
/* K6: Scalar::to_bytes_be (scalar.rs:118-127): n x 32 big-endian bytes, the SHA3 commitment's input */
int arkmpc_scalar_to_bytes_be(arkmpc_ctx* ctx, size_t n, const uint64_t* in, uint8_t* out_bytes);

/* ---- ScalarShare<C> vectors: scalar/share.rs:72-133 via authenticated_scalar.rs batch_* ------ */
int arkmpc_share_add(arkmpc_ctx* ctx, size_t n, const uint64_t* a, const uint64_t* b, uint64_t* out);           /* batch_add :457-489 */
int arkmpc_share_sub(arkmpc_ctx* ctx, size_t n, const uint64_t* a, const uint64_t* b, uint64_t* out);           /* batch_sub :662-688 */
int arkmpc_share_neg(arkmpc_ctx* ctx, size_t n, const uint64_t* a, uint64_t* out);                               /* batch_neg :745-765 */
int arkmpc_share_add_public(arkmpc_ctx* ctx, size_t n, int party_id, const uint64_t mac_key[4],
                            const uint64_t* a, const uint64_t* pub, uint64_t* out);                              /* batch_add_public :493-528 */
int arkmpc_share_sub_public(arkmpc_ctx* ctx, size_t n, int party_id, const uint64_t mac_key[4],
                            const uint64_t* a, const uint64_t* pub, uint64_t* out);                              /* batch_sub_public :691-733 */
int arkmpc_share_mul_public(arkmpc_ctx* ctx, size_t n, const uint64_t* a, const uint64_t* pub, uint64_t* out);  /* batch_mul_public :883-916 */

/* column forms (share / MAC base pointers + element stride in u64 units: 4 = split columns, 8 = AoS view with mac = share + 4) of the
 * public-operand ops, for ScalarShare batches kept in the engine-native split layout; device pointers only */
int arkmpc_share_add_public_v(arkmpc_ctx* ctx, size_t n, int party_id, const uint64_t mac_key[4], const uint64_t* a_share,
                              const uint64_t* a_mac, size_t a_stride, const uint64_t* pub, uint64_t* out_share, uint64_t* out_mac,
                              size_t out_stride);
int arkmpc_share_sub_public_v(arkmpc_ctx* ctx, size_t n, int party_id, const uint64_t mac_key[4], const uint64_t* a_share,
                              const uint64_t* a_mac, size_t a_stride, const uint64_t* pub, uint64_t* out_share, uint64_t* out_mac,
                              size_t out_stride);
int arkmpc_share_mul_public_v(arkmpc_ctx* ctx, size_t n, const uint64_t* a_share, const uint64_t* a_mac, size_t a_stride,
                              const uint64_t* pub, uint64_t* out_share, uint64_t* out_mac, size_t out_stride);
/* Sum for ScalarShare / AuthenticatedScalarResult (share.rs:103-111, authenticated_scalar.rs:563-575): n ScalarShares -> ONE
 * ScalarShare (sum of shares, sum of MACs); n = 0 gives (0, 0) */
int arkmpc_share_sum(arkmpc_ctx* ctx, size_t n, const uint64_t* shares, uint64_t* out_share);
/* layout converters: arkworks AoS ScalarShare records <-> engine-native split columns (n shares, then n MACs).
 * Import once, run the _v entry points on the columns (no dead MAC bytes in K1, cacheable re-reads), export at the end. */
int arkmpc_share_split(arkmpc_ctx* ctx, size_t n, const uint64_t* aos, uint64_t* out_share_col, uint64_t* out_mac_col);
int arkmpc_share_join(arkmpc_ctx* ctx, size_t n, const uint64_t* share_col, const uint64_t* mac_col, uint64_t* out_aos);
/* n copies of one record of `words` u64 (2, 4, ..., 12; `record` is ALWAYS a host pointer): `vec![value; n]` for the
 * constant batches of a preprocessing source (PartyIDBeaverSource, offline_prep.rs:103-170) and fabric constants. */
int arkmpc_fill(arkmpc_ctx* ctx, size_t n, size_t words, const uint64_t* record, uint64_t* out);

/* ---- Beaver multiplication, authenticated_scalar.rs:848-879 -------------------------------- */
/* K1  batch_sub(a,&beaver_a), batch_sub(b,&beaver_b) + the `.share()` projection of open_batch's
 *     network op (:863-868, :141-145).  x,y,a,b: n ScalarShares.  out_de: 2n Scalars, d then e,
 *     exactly the NetworkPayload::ScalarBatch this party sends. */
int arkmpc_beaver_mask(arkmpc_ctx* ctx, size_t n, const uint64_t* x, const uint64_t* y, const uint64_t* a,
                       const uint64_t* b, uint64_t* out_de);
/* K2  open_batch's combine gate (:161-171): out_i = mine_i + peer_i over n Scalars. */
int arkmpc_open_combine(arkmpc_ctx* ctx, size_t n, const uint64_t* mine, const uint64_t* peer, uint64_t* out);
/* K3  de + d[b] + e[a] + [c] (:871-878 == the fused gate :835-840).  d,e: n opened Scalars each. */
int arkmpc_beaver_finish(arkmpc_ctx* ctx, size_t n, int party_id, const uint64_t mac_key[4], const uint64_t* d,
                         const uint64_t* e, const uint64_t* a, const uint64_t* b, const uint64_t* c, uint64_t* out);
/* K2+K3 fused: my_de / peer_de are the two parties' 2n-Scalar d||e buffers from K1. */
int arkmpc_beaver_finish_fused(arkmpc_ctx* ctx, size_t n, int party_id, const uint64_t mac_key[4],
                               const uint64_t* my_de, const uint64_t* peer_de, const uint64_t* a, const uint64_t* b,
                               const uint64_t* c, uint64_t* out);
/* share-view forms: *_share / *_mac column pointers + stride in u64 units (8 = AoS, 4 = split) */
int arkmpc_beaver_mask_v(arkmpc_ctx* ctx, size_t n, const uint64_t* x_share, size_t x_stride, const uint64_t* y_share,
                         size_t y_stride, const uint64_t* a_share, size_t a_stride, const uint64_t* b_share,
                         size_t b_stride, uint64_t* out_de);
int arkmpc_beaver_finish_fused_v(arkmpc_ctx* ctx, size_t n, int party_id, const uint64_t mac_key[4],
                                 const uint64_t* my_de, const uint64_t* peer_de,
                                 const uint64_t* a_share, const uint64_t* a_mac, size_t a_stride,
                                 const uint64_t* b_share, const uint64_t* b_mac, size_t b_stride,
                                 const uint64_t* c_share, const uint64_t* c_mac, size_t c_stride,
                                 uint64_t* out_share, uint64_t* out_mac, size_t out_stride);

/* range forms: gates [lo, lo + n) of a larger batch of N.  d and e are addressed separately, so a shard writes its slice of the FULL
 * d||e buffer (out_d = full + 4 lo, out_e = full + 4 (N + lo)) -- on its own device or, with peer access, in another device's memory --
 * and K2+K3 reads the two parties' d / e slices from wherever they lie.  Device pointers only. */
int arkmpc_beaver_mask_to(arkmpc_ctx* ctx, size_t n, const uint64_t* x_share, size_t x_stride, const uint64_t* y_share,
                          size_t y_stride, const uint64_t* a_share, size_t a_stride, const uint64_t* b_share, size_t b_stride,
                          uint64_t* out_d, uint64_t* out_e);
/* K1 writing its payload twice (both 2n Scalars, d then e): out_de stays with the party for its own K2+K3, out_de_msg is the message a
 * device-resident link hands to the peer (network/mock.rs moves payloads; the sender must not alias a buffer it still reads).  Replaces
 * K1 + a device-to-device copy: one launch less in every round's dependent chain. */
int arkmpc_beaver_mask_dup(arkmpc_ctx* ctx, size_t n, const uint64_t* x_share, size_t x_stride, const uint64_t* y_share,
                           size_t y_stride, const uint64_t* a_share, size_t a_stride, const uint64_t* b_share, size_t b_stride,
                           uint64_t* out_de, uint64_t* out_de_msg);
int arkmpc_beaver_finish_fused_from(arkmpc_ctx* ctx, size_t n, int party_id, const uint64_t mac_key[4],
                                    const uint64_t* my_d, const uint64_t* my_e, const uint64_t* peer_d, const uint64_t* peer_e,
                                    const uint64_t* a_share, const uint64_t* a_mac, size_t a_stride,
                                    const uint64_t* b_share, const uint64_t* b_mac, size_t b_stride,
                                    const uint64_t* c_share, const uint64_t* c_mac, size_t c_stride,
                                    uint64_t* out_share, uint64_t* out_mac, size_t out_stride);

/* ---- streaming host-to-host forms (csrc/arkmpc_stream.inc) -----------------------------------------------------------------------
 * For a caller whose operands ARE host memory -- the `Vec<ScalarShare<C>>`s a gate closure receives (fabric.rs:841-854) -- and who
 * wants host vectors back: what benches/batch_ops.rs:19-39 times.  Over PCIe the path is bound by the host link (384 B up, 128 B down
 * per party-gate), so these entry points run a three-stream pipeline (upload DMA | kernels | download DMA) that keeps the upload
 * direction busy from the first byte to the last and hides the kernels and the downloads under it.  They take HOST pointers whatever
 * the context's buffer mode, any alignment a Rust Vec has (8 bytes).  Pageable buffers are pinned in place for the duration of the call
 * (hipHostRegister) so that the copies are true DMAs.  A caller that keeps its vectors in pinned memory -- arkmpc_host_alloc, or arkmpc_host_register ONCE for vectors it
 * keeps -- gets the fast form: a phase whose vectors are all pinned (and 16-byte aligned) runs with NO copies -- one kernel reads the
 * records where they lie in host memory and writes the payload / result vector in place (7.3 instead of 8.1 ms per 2^20 gates, 0.97 of
 * the link).  Same words either way; sessions with several open on one context (both parties of an in-process run) are supported.
 * A pageable buffer is registered in place for the duration of the call and moved by DMA (8.2 ms per 2^20 gates); NO KERNEL addresses a vector
 * the library registered itself: on this platform such a kernel can read stale memory when the vector's address had an earlier registered life
 * (freed, handed out again by malloc with other physical pages) -- DESIGN section 4.  Kernels address in place only what the CALLER holds in
 * pinned memory: arkmpc_host_alloc, or arkmpc_host_register ONCE for a vector it keeps (register once, not per gate).  ARKMPC_PIN_IN_PLACE=0
 * never registers (the runtime's pageable copies, about half the rate). */
int arkmpc_host_register(void* ptr, size_t bytes);      /* already registered = ARKMPC_OK (ARKMPC_ERR_BAD_ARG if the same pointer is registered again with a LARGER size: unregister first) */
int arkmpc_host_unregister(void* ptr);                  /* drops what arkmpc_host_register took; memory pinned by somebody else is left alone */
/* FIRST-LIFE RULE (round 6): every registration made through this library -- the per-call pins of pageable vectors and arkmpc_host_register --
 * is tracked process-wide; when it ends its pages are RETIRED, and a vector registered later over retired addresses (a fresh Vec per gate gets
 * recycled addresses from the allocator) is never addressed by a kernel: it takes the DMA pipeline, same words, and is counted in
 * arkmpc_ctx_stats.zc_refused_reused_address.  So: register long-lived memory ONCE (or use arkmpc_host_alloc, whose blocks are recycled without
 * ever being unregistered), and arkmpc_host_unregister BEFORE freeing -- freeing registered memory is undefined for the HIP runtime too.
 * Memory the caller pins with the runtime directly (hipHostRegister, hipHostMalloc, torch's pinned tensors) is outside this bookkeeping: it is
 * addressed in place, and a caller that re-registers recycled addresses that way owns the rule itself. */
int arkmpc_host_alloc(size_t bytes, void** out_ptr);    /* pinned allocation (hipHostMalloc), RECYCLED: a freed block comes back from a free list by size */
int arkmpc_host_free(void* ptr);                        /* class in microseconds (the runtime's own alloc + free of 64 MiB cost 16 ms); contents are not cleared */
int arkmpc_host_trim(void);                             /* returns the free list (at most ARKMPC_HOST_POOL_MB, default 4096 MiB) to the runtime */
/* AuthenticatedScalarResult::batch_mul (authenticated_scalar.rs:848-879) as a two-phase session around the d||e exchange:
 *   _begin   x, y, a, b, c: n ScalarShares each (arkworks records); out_de: 2n Scalars, d then e -- the ScalarBatch this party sends
 *            (:863-868, :141-145).  Enqueues the uploads, K1 chunk by chunk and the payload downloads, and returns.
 *   _poll_de / _wait_de   how many leading gates of d AND e have landed in out_de (never blocks) / block until all have: the sender
 *            of a real link can start transmitting before the batch is complete.
 *   _finish  peer_de: the 2n Scalars received (:871-878); out: n ScalarShares.  Blocks until `out` is complete; ends the session (also
 *            when it returns an error).  _abort ends a session without phase 2.
 * out_de is the caller's again once _wait_de has returned (it has been sent; it may be freed or reused before _finish).
 * The input vectors must stay valid, unmodified and (if the caller pinned them) pinned until _finish / _abort returns; sessions must be
 * ended before their context is destroyed.  Sessions of different contexts (rayon workers) are independent; two parties sharing one GPU
 * do best as two sessions of ONE context driven by one thread (their uploads then queue instead of racing on the link). */
typedef struct arkmpc_hostmul arkmpc_hostmul;
int arkmpc_hostmul_begin(arkmpc_ctx* ctx, size_t n, const uint64_t* x, const uint64_t* y, const uint64_t* a, const uint64_t* b,
                         const uint64_t* c, uint64_t* out_de, arkmpc_hostmul** out_session);
int arkmpc_hostmul_poll_de(arkmpc_hostmul* session, size_t* out_gates);
int arkmpc_hostmul_wait_de(arkmpc_hostmul* session);
int arkmpc_hostmul_finish(arkmpc_hostmul* session, int party_id, const uint64_t mac_key[4], const uint64_t* peer_de, uint64_t* out);
int arkmpc_hostmul_abort(arkmpc_hostmul* session);
/* Placement and range forms of the same session.
 * PLACEMENT: every vector of a session is looked at on its own and may be pageable host memory, pinned host memory or DEVICE memory of the
 *   context's GPU (16-byte aligned; memory of another GPU is ARKMPC_ERR_BAD_ARG).  A circuit keeps x, y -- the previous gates' outputs -- and
 *   its result resident in HBM while the triples a, b, c lie in the preprocessing source's host memory (fabric.rs:894-915,
 *   offline_prep.rs:65-81) and the payloads cross a host link: resident vectors are read / written where they are, nothing is staged for them.
 * RANGE: _begin_range / _finish_async address the two halves of a d||e vector separately, so a session can be gates [lo, lo + n) of a batch
 *   of N: pass x + 8 lo ... c + 8 lo, out_d = de + 4 lo, out_e = de + 4 (N + lo) (same for peer_d / peer_e, out + 8 lo).  This is what the
 *   multi-device group's sessions below are made of.  flags: ARKMPC_HOSTMUL_NO_PIN = the session registers no host memory itself (the caller
 *   pins whole vectors once for all of its range sessions; unpinned vectors then travel as the runtime's pageable copies).
 * _finish_async enqueues phase 2 and returns; _end blocks until `out` is complete and ends the session (arkmpc_hostmul_finish = the two
 *   together).  After a failed _finish_async the session is still open: end it with _end or _abort.  Several sessions -- one per GPU of a
 *   group -- therefore run their phases concurrently from one host thread. */
#define ARKMPC_HOSTMUL_NO_PIN 1u
int arkmpc_hostmul_begin_range(arkmpc_ctx* ctx, size_t n, const uint64_t* x, const uint64_t* y, const uint64_t* a, const uint64_t* b,
                               const uint64_t* c, uint64_t* out_d, uint64_t* out_e, unsigned flags, arkmpc_hostmul** out_session);
int arkmpc_hostmul_finish_async(arkmpc_hostmul* session, int party_id, const uint64_t mac_key[4], const uint64_t* peer_d,
                                const uint64_t* peer_e, uint64_t* out);
int arkmpc_hostmul_end(arkmpc_hostmul* session);
/* The same session with the payloads in their WIRE form -- the frames QuicTwoPartyNet writes and reads (network/quic.rs:226-251, :303-306;
 * format below under "Wire format"), ~115 text bytes per scalar, rendered and parsed on the GPU so that the host never runs serde_json over
 * 2n scalars (where the QUIC path's time goes):
 *   _begin_wire   phase 1, then NetworkOutbound{result_id, ScalarBatch(d||e)} as a frame into out_frame (capacity >= arkmpc_wire_frame_bound(2 n));
 *                 *out_len = its length.  Blocks until the frame is complete.
 *   _finish_wire  peer_frame = the frame received; it must be a ScalarBatch of exactly 2n canonical scalars (anything else: ARKMPC_ERR_BAD_ARG, as
 *                 serde_json / deserialize_uncompressed would fail); *out_result_id (may be NULL) = its result_id.  Then phase 2 as _finish. */
int arkmpc_hostmul_begin_wire(arkmpc_ctx* ctx, size_t n, const uint64_t* x, const uint64_t* y, const uint64_t* a, const uint64_t* b,
                              const uint64_t* c, uint64_t result_id, uint8_t* out_frame, size_t out_cap, size_t* out_len,
                              arkmpc_hostmul** out_session);
int arkmpc_hostmul_finish_wire(arkmpc_hostmul* session, int party_id, const uint64_t mac_key[4], const uint8_t* peer_frame, size_t peer_len,
                               uint64_t* out, uint64_t* out_result_id);

/* ---- batch open + MAC check, authenticated_scalar.rs:278-354 ------------------------------- */
/* the `.share()` projection sent by open_batch (:141-145): n ScalarShares -> n Scalars */
int arkmpc_share_extract(arkmpc_ctx* ctx, size_t n, const uint64_t* shares, uint64_t* out_share_values);
/* K4  mac_key * value - share.mac() (:299-311) */
int arkmpc_mac_check_shares(arkmpc_ctx* ctx, size_t n, const uint64_t mac_key[4], const uint64_t* opened,
                            const uint64_t* shares, uint64_t* out_chk);
/* K2+K4 fused: opened_i = shares_i.share + peer_i ; chk_i = mac_key*opened_i - shares_i.mac */
int arkmpc_open_and_mac_check(arkmpc_ctx* ctx, size_t n, const uint64_t mac_key[4], const uint64_t* shares,
                              const uint64_t* peer_share_values, uint64_t* out_opened, uint64_t* out_chk);
/* K2+K4 on share / MAC columns (stride in u64 units: 4 = split columns, 8 = AoS view with mac = share + 4).  With split columns the
 * `.share()` payload a party sends IS its share column: no arkmpc_share_extract pass, and the MAC half is read once. */
int arkmpc_open_and_mac_check_v(arkmpc_ctx* ctx, size_t n, const uint64_t mac_key[4], const uint64_t* share_col, const uint64_t* mac_col,
                                size_t stride, const uint64_t* peer_share_values, uint64_t* out_opened, uint64_t* out_chk);
/* K5  all(mine_i + peer_i == 0) (:218-219).  Blocking; *out_ok = 1 or 0. */
int arkmpc_mac_verify(arkmpc_ctx* ctx, size_t n, const uint64_t* mine, const uint64_t* peer, int* out_ok);
/* Non-blocking form for callers that verify several ranges (sharded batches, the two parties of a mock run): _async enqueues K5
 * on the context's stream and returns; failures accumulate in a per-context sticky flag.  _result waits for the stream, reports
 * 1 iff NO range verified since the last collection failed, and clears the flag.  (arkmpc_mac_verify = _async + _result.) */
int arkmpc_mac_verify_async(arkmpc_ctx* ctx, size_t n, const uint64_t* mine, const uint64_t* peer);
int arkmpc_mac_verify_result(arkmpc_ctx* ctx, int* out_ok);
/* H1  HashCommitmentResult::batch_commit / HashCommitment::verify (commitment.rs:63-89, :30-43):
 *     out = from_be_bytes_mod_order(SHA3-256(BE(v_0)||...||BE(v_{n-1})||BE(blinder))).
 *     K6 runs on the GPU, the sponge on the host (sequential by definition), overlapped with D2H.
 *     Blocking; `blinder` and `out_commitment` are host pointers. */
int arkmpc_commit_sha3(arkmpc_ctx* ctx, size_t n, const uint64_t* values, const uint64_t blinder[4],
                       uint64_t out_commitment[4]);
/* plain SHA3-256 of a host buffer (the `sha3` crate's Sha3_256) */
int arkmpc_sha3_256(const uint8_t* msg, size_t len, uint8_t out32[32]);
/* which Keccak-f[1600] absorb loop this process uses for the sponge: "portable" | "scalar" | "bmi" | "avx512" | "lanes" | "rows" -- the one
 * ARKMPC_KECCAK names if this CPU supports it, else the fastest of a timed trial at first use (csrc/sha3_host.hip).  Static string. */
const char* arkmpc_sha3_loop(void);

/* ---- BN254 G1 points (context field must be ARKMPC_BN254_FR) -------------------------------- */
int arkmpc_g1_add(arkmpc_ctx* ctx, size_t n, const uint64_t* a, const uint64_t* b, uint64_t* out);  /* curve.rs:203-209 */
int arkmpc_g1_sub(arkmpc_ctx* ctx, size_t n, const uint64_t* a, const uint64_t* b, uint64_t* out);  /* curve.rs:282-288 */
int arkmpc_g1_neg(arkmpc_ctx* ctx, size_t n, const uint64_t* a, uint64_t* out);                      /* curve.rs:367-373 */
/* K8  CurvePoint * Scalar (curve.rs:403-409); CurvePointResult::batch_mul (curve.rs:459-479) */
int arkmpc_g1_scalar_mul(arkmpc_ctx* ctx, size_t n, const uint64_t* points, const uint64_t* scalars, uint64_t* out);
/* generator * scalar_i */
int arkmpc_g1_generator_mul(arkmpc_ctx* ctx, size_t n, const uint64_t* scalars, uint64_t* out);
/* normalise to affine: out_xy = n x { x[4], y[4] } Montgomery Fq; out_inf[i] = 1 for the identity */
int arkmpc_g1_to_affine(arkmpc_ctx* ctx, size_t n, const uint64_t* points, uint64_t* out_xy, uint8_t* out_inf);
/* CurvePoint::to_bytes (curve.rs:103-108): arkworks compressed encoding, n x 32 bytes */
int arkmpc_g1_to_bytes(arkmpc_ctx* ctx, size_t n, const uint64_t* points, uint8_t* out_bytes);
/* CurvePoint::from_bytes (curve.rs:110-114) = deserialize_compressed with validation: n x 32 bytes -> n points (x, y, 1)
 * (identity as (1,1,0)); out_ok[i] = 1 if the bytes are a point encoding (x < q, at most one flag bit, x^3 + 3 a square),
 * else 0 and the identity is stored.  What a party runs on a received PointBatch (authenticated_curve.rs:74-89). */
int arkmpc_g1_from_bytes(arkmpc_ctx* ctx, size_t n, const uint8_t* bytes, uint64_t* out_points, uint8_t* out_ok);
/* PointShare ops, curve/share.rs:55-114 via authenticated_curve.rs batch_* */
int arkmpc_pointshare_add(arkmpc_ctx* ctx, size_t n, const uint64_t* a, const uint64_t* b, uint64_t* out);          /* :396-426 */
int arkmpc_pointshare_sub(arkmpc_ctx* ctx, size_t n, const uint64_t* a, const uint64_t* b, uint64_t* out);          /* :520-550 */
int arkmpc_pointshare_neg(arkmpc_ctx* ctx, size_t n, const uint64_t* a, uint64_t* out);                              /* :604-621 */
int arkmpc_pointshare_mul_public(arkmpc_ctx* ctx, size_t n, const uint64_t* shares, const uint64_t* scalars,
                                 uint64_t* out);                                                                     /* :718-751 */
int arkmpc_pointshare_add_public(arkmpc_ctx* ctx, size_t n, int party_id, const uint64_t mac_key[4],
                                 const uint64_t* shares, const uint64_t* pub_points, uint64_t* out);                 /* :429-463 */
/* PointShare::sub_public = add_public(-rhs) (curve/share.rs:63-65; the `AuthenticatedPointResult - CurvePointResult` operators,
 * authenticated_curve.rs:553-575) */
int arkmpc_pointshare_sub_public(arkmpc_ctx* ctx, size_t n, int party_id, const uint64_t mac_key[4],
                                 const uint64_t* shares, const uint64_t* pub_points, uint64_t* out);
int arkmpc_scalarshare_mul_generator(arkmpc_ctx* ctx, size_t n, const uint64_t* scalar_shares, uint64_t* out);      /* :754-780 */
/* The point-side K3 as ONE gate: AuthenticatedPointResult::batch_mul after the openings (authenticated_curve.rs:703-713).  Given the opened
 * d = x - a (n Scalars), the opened eG = yG - [b]G (n points) and this party's triple shares a, b, c (n ScalarShares each),
 *     out = ([a] + d) * eG + ([c] + d [b]) * G
 * which equals the reference's deG + d[bG] + [a]eG + [c]G share by share and MAC by MAC ([bG] = [b]G; "[a] + d" is
 * ScalarShare::add_public, share.rs:74-77): 2 variable-base + 2 generator scalar-muls per element instead of 6 + 2.
 * out: n PointShares.  arkmpc_edpoint_beaver_finish is the same gate on a CURVE25519_FR context (16-word points).
 * A composite of six entry points, each taking the context lock on its own: calls of other threads on the SAME context may interleave
 * between them (harmless -- they are ordered on the one stream -- but the gate is not atomic with respect to them). */
int arkmpc_point_beaver_finish(arkmpc_ctx* ctx, size_t n, int party_id, const uint64_t mac_key[4], const uint64_t* d_open,
                               const uint64_t* eG_open, const uint64_t* a, const uint64_t* b, const uint64_t* c, uint64_t* out_shares);
int arkmpc_edpoint_beaver_finish(arkmpc_ctx* ctx, size_t n, int party_id, const uint64_t mac_key[4], const uint64_t* d_open,
                                 const uint64_t* eG_open, const uint64_t* a, const uint64_t* b, const uint64_t* c, uint64_t* out_shares);
int arkmpc_scalarshare_mul_point(arkmpc_ctx* ctx, size_t n, const uint64_t* scalar_shares, const uint64_t* points,
                                 uint64_t* out);                                                                     /* curve.rs:483-517 */
/* sums: the reduction gate of AuthenticatedPointResult::msm (authenticated_curve.rs:796-805) / PointShare::sum
 * (curve/share.rs:85-92).  n = 0 gives the identity.  out: ONE point (12 x u64) / ONE PointShare (24 x u64). */
int arkmpc_g1_sum(arkmpc_ctx* ctx, size_t n, const uint64_t* points, uint64_t* out_point);
int arkmpc_pointshare_sum(arkmpc_ctx* ctx, size_t n, const uint64_t* shares, uint64_t* out_share);
/* Variable-base multi-scalar multiplication, sum_i scalars[i] * points[i]  (CurvePoint::msm, curve.rs:549-560, which
 * batch-normalises to affine and calls ark-ec's VariableBaseMSM; also the gate body of msm_results, :588-603 / :661-684).
 * points: n Jacobian points (12 x u64), any representative, identities allowed; scalars: n x 4 u64 Montgomery.
 * out: ONE point.  n = 0 gives the identity.  Bucket (Pippenger) method on the device; the summation order is fixed, so
 * the output representative is deterministic. */
int arkmpc_g1_msm(arkmpc_ctx* ctx, size_t n, const uint64_t* points, const uint64_t* scalars, uint64_t* out_point);
/* CurvePointResult::msm_authenticated (curve.rs:618-642 / :701-731): authenticated scalars x public points ->
 * PointShare(msm(shares, points), msm(macs, points)).  scalar_shares: n ScalarShares (8 x u64); out: ONE PointShare
 * (24 x u64).  Both columns go through one sort / one set of launches over the shared affine points. */
int arkmpc_g1_msm_authenticated(arkmpc_ctx* ctx, size_t n, const uint64_t* points, const uint64_t* scalar_shares,
                                uint64_t* out_share);
/* the `.share()` projection of AuthenticatedPointResult::open_batch (:74-89): n PointShares -> n points */
int arkmpc_pointshare_extract(arkmpc_ctx* ctx, size_t n, const uint64_t* shares, uint64_t* out_points);
/* value * mac_key - share.mac() per element (authenticated_curve.rs:215-220) */
int arkmpc_point_mac_check_shares(arkmpc_ctx* ctx, size_t n, const uint64_t mac_key[4], const uint64_t* opened_points,
                                  const uint64_t* shares, uint64_t* out_chk_points);
/* K9  per-element HashCommitmentResult::commit on points (authenticated_curve.rs:227 -> commitment.rs:58-89):
 *     out_i = from_be_bytes_mod_order(SHA3-256(to_bytes(points_i) || to_bytes_be(blinders_i))).  n independent one-block
 *     hashes, so unlike arkmpc_commit_sha3 this runs entirely on the GPU.  blinders / out: n Scalars (same space as points). */
int arkmpc_commit_points_sha3(arkmpc_ctx* ctx, size_t n, const uint64_t* points, const uint64_t* blinders,
                              uint64_t* out_commitments);
/* all(mine_i + peer_i == identity) (authenticated_curve.rs:127-131), per element: out_ok[i] in {0,1} (host or device per mode) */
int arkmpc_point_mac_verify(arkmpc_ctx* ctx, size_t n, const uint64_t* mine, const uint64_t* peer, uint8_t* out_ok);

/* ---- Curve25519 points (context field must be ARKMPC_CURVE25519_FR) -------------------------------------------------
 * ark_curve25519::EdwardsProjective (README.md:24): twisted-Edwards extended coordinates { x[4], y[4], t[4], z[4] } over
 * 2^255 - 19 in Montgomery form = 16 x u64, identity (0, 1, 0, 1); PointShare = 32 x u64.  Same CurvePoint / PointShare
 * semantics as the BN254 entry points above (curve.rs:203-409, curve/share.rs:55-114). */
int arkmpc_ed_add(arkmpc_ctx* ctx, size_t n, const uint64_t* a, const uint64_t* b, uint64_t* out);
int arkmpc_ed_sub(arkmpc_ctx* ctx, size_t n, const uint64_t* a, const uint64_t* b, uint64_t* out);
int arkmpc_ed_neg(arkmpc_ctx* ctx, size_t n, const uint64_t* a, uint64_t* out);
int arkmpc_ed_scalar_mul(arkmpc_ctx* ctx, size_t n, const uint64_t* points, const uint64_t* scalars, uint64_t* out);
int arkmpc_ed_generator_mul(arkmpc_ctx* ctx, size_t n, const uint64_t* scalars, uint64_t* out);
int arkmpc_ed_to_affine(arkmpc_ctx* ctx, size_t n, const uint64_t* points, uint64_t* out_xy);        /* n x { x[4], y[4] } */
int arkmpc_ed_to_bytes(arkmpc_ctx* ctx, size_t n, const uint64_t* points, uint8_t* out_bytes);         /* y LE, bit 7 = x > -x */
/* CurvePoint::from_bytes on Curve25519 (curve.rs:110-114 -> twisted-Edwards deserialize_compressed with validation): n x 32
 * bytes -> n extended points (x, y, xy, 1); out_ok[i] = 0 (and the identity) unless y < q, x^2 = (y^2-1)/(d y^2+1) is a
 * square and the point lies in the prime-order subgroup (cofactor 8: [l]P = O). */
int arkmpc_ed_from_bytes(arkmpc_ctx* ctx, size_t n, const uint8_t* bytes, uint64_t* out_points, uint8_t* out_ok);
int arkmpc_edshare_add(arkmpc_ctx* ctx, size_t n, const uint64_t* a, const uint64_t* b, uint64_t* out);
int arkmpc_edshare_sub(arkmpc_ctx* ctx, size_t n, const uint64_t* a, const uint64_t* b, uint64_t* out);
int arkmpc_edshare_neg(arkmpc_ctx* ctx, size_t n, const uint64_t* a, uint64_t* out);
int arkmpc_edshare_mul_public(arkmpc_ctx* ctx, size_t n, const uint64_t* shares, const uint64_t* scalars, uint64_t* out);
int arkmpc_edshare_add_public(arkmpc_ctx* ctx, size_t n, int party_id, const uint64_t mac_key[4], const uint64_t* shares,
                              const uint64_t* pub_points, uint64_t* out);
int arkmpc_edshare_sub_public(arkmpc_ctx* ctx, size_t n, int party_id, const uint64_t mac_key[4], const uint64_t* shares,
                              const uint64_t* pub_points, uint64_t* out);                                        /* curve/share.rs:63-65 */
int arkmpc_scalarshare_mul_ed_generator(arkmpc_ctx* ctx, size_t n, const uint64_t* scalar_shares, uint64_t* out);
int arkmpc_scalarshare_mul_ed_point(arkmpc_ctx* ctx, size_t n, const uint64_t* scalar_shares, const uint64_t* points,
                                    uint64_t* out);                                                                /* curve.rs:483-517 */
/* The authenticated-point protocol pieces on Curve25519 (the reference is generic over C: authenticated_curve.rs:66-283, 682-806),
 * same semantics as arkmpc_pointshare_extract / point_mac_check_shares / point_mac_verify / commit_points_sha3 / g1_sum /
 * pointshare_sum above, on 16 / 32 x u64 elements and the twisted-Edwards compressed encoding. */
int arkmpc_edshare_extract(arkmpc_ctx* ctx, size_t n, const uint64_t* shares, uint64_t* out_points);
int arkmpc_ed_mac_check_shares(arkmpc_ctx* ctx, size_t n, const uint64_t mac_key[4], const uint64_t* opened_points,
                               const uint64_t* shares, uint64_t* out_chk_points);
int arkmpc_ed_mac_verify(arkmpc_ctx* ctx, size_t n, const uint64_t* mine, const uint64_t* peer, uint8_t* out_ok);
int arkmpc_commit_ed_points_sha3(arkmpc_ctx* ctx, size_t n, const uint64_t* points, const uint64_t* blinders,
                                 uint64_t* out_commitments);
/* CurvePoint::msm / msm_authenticated on Curve25519 (curve.rs:549-560, :618-642 -- generic over C): the bucket method on the complete
 * twisted-Edwards addition (no exceptional lanes, no affine conversion pass); semantics of arkmpc_g1_msm / arkmpc_g1_msm_authenticated
 * on 16-word points, out = ONE point / ONE PointShare (32 x u64); n = 0 gives the identity */
int arkmpc_ed_msm(arkmpc_ctx* ctx, size_t n, const uint64_t* points, const uint64_t* scalars, uint64_t* out_point);
int arkmpc_ed_msm_authenticated(arkmpc_ctx* ctx, size_t n, const uint64_t* points, const uint64_t* scalar_shares, uint64_t* out_share);
int arkmpc_ed_sum(arkmpc_ctx* ctx, size_t n, const uint64_t* points, uint64_t* out_point);
int arkmpc_edshare_sum(arkmpc_ctx* ctx, size_t n, const uint64_t* shares, uint64_t* out_share);

/* ---------------------------------------------------------------------------------------------------------------------
 * Wire format of the batches that cross the party-to-party link (csrc/arkmpc_wire.hip).
 * A frame is what QuicTwoPartyNet writes (network/quic.rs:303-306): u64 little-endian length, then
 * serde_json::to_vec(&NetworkOutbound{result_id, payload}) (network.rs:33-60), i.e. the compact text
 *   {"result_id":<id>,"payload":{"ScalarBatch":[[b0,...,b31],...]}}
 * with every Scalar as its 32 canonical little-endian bytes (scalar.rs:186-192) and every CurvePoint as its 32
 * compressed bytes (curve.rs:50-55, :103-108; variant "PointBatch"), each byte a decimal number.
 * Frame pointers follow the context's buffer mode like every other buffer.  Encoders and decoders block (the frame
 * length / element count is data dependent).  Decoders validate the whole grammar: a byte > 255, a scalar >= the modulus or
 * any syntax error returns ARKMPC_ERR_BAD_ARG (serde_json / deserialize_uncompressed would return an error to the caller,
 * scalar.rs:195-201).  A frame that is not in serde_json::to_vec's compact form (what a reference peer sends and the GPU parser reads
 * directly) is re-read on the host with the semantics serde_json::from_slice gives a derived struct (network/quic.rs:233-251): JSON
 * whitespace between tokens, the two fields in any order, unknown fields skipped whatever value they hold, escaped keys compared after
 * unescaping, a known field given twice = error, the payload an object with exactly one variant key, nothing but whitespace after the
 * closing brace, nesting limited to 128 levels; also the two-element SEQUENCE form [result_id, payload] a derived struct deserialises from
 * (fields in declaration order, network.rs:33-40; one or three elements = error) -- and, if it is a message, rewritten to the compact form and parsed on the GPU.
 * A device-mode frame buffer needs no padding: no byte at or beyond frame_len is read. */
#define ARKMPC_WIRE_SCALAR_BATCH 0
#define ARKMPC_WIRE_POINT_BATCH 1
/* capacity (bytes) an encode call needs for n elements: 8 + 80 + 131 n + 3 */
int arkmpc_wire_frame_bound(size_t n, size_t* out_bytes);
/* NetworkPayload::ScalarBatch of n Montgomery scalars (what open_batch / the Beaver d||e exchange send,
 * authenticated_scalar.rs:141-145).  *out_len = bytes written (length prefix included). */
int arkmpc_wire_encode_scalar_batch(arkmpc_ctx* ctx, uint64_t result_id, size_t n, const uint64_t* scalars, uint8_t* out_frame,
                                    size_t out_cap, size_t* out_len);
/* n ready-made 32-byte records under the given variant (PointBatch: the output of arkmpc_g1_to_bytes / arkmpc_ed_to_bytes) */
int arkmpc_wire_encode_bytes32(arkmpc_ctx* ctx, int kind, uint64_t result_id, size_t n, const uint8_t* records, uint8_t* out_frame,
                               size_t out_cap, size_t* out_len);
/* Parse a received ScalarBatch frame into Montgomery scalars.  max_n = capacity of out_scalars (elements); *out_n = count. */
int arkmpc_wire_decode_scalar_batch(arkmpc_ctx* ctx, const uint8_t* frame, size_t frame_len, size_t max_n, uint64_t* out_scalars,
                                    size_t* out_n, uint64_t* out_result_id);
/* Parse a ScalarBatch / PointBatch frame into raw 32-byte records; *out_kind = ARKMPC_WIRE_* */
int arkmpc_wire_decode_bytes32(arkmpc_ctx* ctx, const uint8_t* frame, size_t frame_len, size_t max_n, uint8_t* out_records, size_t* out_n,
                               uint64_t* out_result_id, int* out_kind);

/* ---------------------------------------------------------------------------------------------------------------------
 * Multi-device group: ONE process drives N GPUs of one node (csrc/arkmpc_group.hip).
 * A party of the reference is one process (MpcFabric::new, fabric.rs:402-466), so the 8-GPU form of the path has to live behind
 * the FFI: a group owns one context (device + stream) per member and range-shards every batch of n independent gates as
 * member g <- [g*n/G, (g+1)*n/G) (authenticated_scalar.rs:677-687, :904-913: element i depends on element i only).  No collective
 * in the arithmetic.  Device ids may repeat (members then share a GPU): that is how the path is tested on a one-GPU box.
 *
 * A SHARDED VECTOR is an array of G device pointers; member g's pointer addresses `segs` consecutive segments of cnt_g elements on
 * ITS device:  Scalars: 1 segment of 4 words;  d||e: 2 segments of 4 words (d_g then e_g);  ScalarShares in ARKMPC_LAYOUT_AOS:
 * 1 segment of 8 words;  in ARKMPC_LAYOUT_SPLIT: 2 segments of 4 words (share column, then MAC column).  The corresponding FULL
 * vector (host or one device) is `segs` segments of n elements.
 *
 * Ordering: group calls enqueue on the members' streams and return (unless noted "blocks"); work of one member is ordered by its
 * stream, cross-member movement by events the group inserts.  Group calls are serialised by a group mutex; the member contexts
 * (arkmpc_group_ctx) remain usable with every single-device entry point above. */
typedef struct arkmpc_group arkmpc_group;
int arkmpc_group_create(int field_id, int n_devices, const int* device_ids, arkmpc_group** out_group);
int arkmpc_group_destroy(arkmpc_group* grp);
int arkmpc_group_size(const arkmpc_group* grp);
int arkmpc_group_device(const arkmpc_group* grp, int member);
arkmpc_ctx* arkmpc_group_ctx(arkmpc_group* grp, int member);
/* member's index range of a batch of n: lo = member*n/G, count = (member+1)*n/G - lo (= ark-mpc_amd/sharding.py shard_range) */
int arkmpc_group_shard_range(const arkmpc_group* grp, size_t n, int member, size_t* out_lo, size_t* out_count);
/* 1 if member `from` can address member `to`'s memory (same device, or xGMI peer mapping enabled at group creation) */
int arkmpc_group_peer_access(const arkmpc_group* grp, int from_member, int to_member);
int arkmpc_group_sync(arkmpc_group* grp);                       /* blocks until every member's stream has drained */
/* device-side ordering between two groups on the same devices: member m of `grp` waits for what member m of `producer` has submitted so far.
 * REQUIRED between two in-process parties that hand each other shard pointers: call it on the consumer after the producer's K1 (its K2+K3 reads
 * the producer's d||e shards) and on the producer's group after the consumer's K2+K3 before the next K1 overwrites those shards -- the two
 * groups' streams are otherwise unordered. */
int arkmpc_group_wait_group(arkmpc_group* grp, arkmpc_group* producer);
const char* arkmpc_group_last_error(arkmpc_group* grp);
/* sharded storage: out_shards / shards = arrays of G pointers */
int arkmpc_group_malloc(arkmpc_group* grp, size_t n, size_t segs, size_t elem_words, uint64_t** out_shards);
int arkmpc_group_free(arkmpc_group* grp, uint64_t* const* shards);
/* host full vector <-> shards: the host vector is pinned once (unless the caller already did), every member's DMA is enqueued on its own
 * stream / PCIe link before any is waited for; blocks until all have landed */
int arkmpc_group_scatter_h2d(arkmpc_group* grp, size_t n, size_t segs, size_t elem_words, const uint64_t* host, uint64_t* const* shards);
int arkmpc_group_gather_d2h(arkmpc_group* grp, size_t n, size_t segs, size_t elem_words, const uint64_t* const* shards, uint64_t* host);
/* host Vec<ScalarShare> (n arkworks records) <-> sharded ScalarShare vector in `layout`; blocks.  _from_host into ARKMPC_LAYOUT_SPLIT reads the
 * pinned records in place, one kernel per member over its own link writing both columns (no staging, no split pass) */
int arkmpc_group_shares_from_host(arkmpc_group* grp, int layout, size_t n, const uint64_t* host_records, uint64_t* const* shards);
int arkmpc_group_shares_to_host(arkmpc_group* grp, int layout, size_t n, const uint64_t* const* shards, uint64_t* host_records);
/* AuthenticatedScalarResult::batch_mul (authenticated_scalar.rs:848-879) as a streaming session over the GROUP: the arkmpc_hostmul_* session
 * above for a party that owns several GPUs and whose operands are host vectors (benches/batch_ops.rs:19-39).  Same arguments, same words: x, y,
 * a, b, c = n ScalarShare records each in HOST memory, out_de / peer_de = 2n Scalars (d then e), out = n records.  Member g runs gates
 * [g n/G, (g+1) n/G) of the same vectors as a range session on its own device and PCIe link; vectors from arkmpc_host_alloc /
 * arkmpc_host_register are read and written in place -- the form that reaches the links' rate -- (pageable ones are registered once per
 * call, whole, and every member moves its range by DMA), and because the member
 * calls only enqueue, all G links are busy together from one host thread.  Host-fed, a party is link-bound 20x below the kernels' rate, so the
 * links are what more GPUs add.  _begin returns once phase 1 is enqueued on every member; _poll_de = leading gates of d AND e complete in out_de
 * (members in range order); _wait_de blocks until all of out_de is (it is the caller's again afterwards); _finish blocks until `out` is
 * complete and ends the session (also on error); _abort ends it without phase 2. */
typedef struct arkmpc_group_hostmul arkmpc_group_hostmul;
int arkmpc_group_hostmul_begin(arkmpc_group* grp, size_t n, const uint64_t* x, const uint64_t* y, const uint64_t* a, const uint64_t* b,
                               const uint64_t* c, uint64_t* out_de, arkmpc_group_hostmul** out_session);
int arkmpc_group_hostmul_poll_de(arkmpc_group_hostmul* session, size_t* out_gates);
int arkmpc_group_hostmul_wait_de(arkmpc_group_hostmul* session);
int arkmpc_group_hostmul_finish(arkmpc_group_hostmul* session, int party_id, const uint64_t mac_key[4], const uint64_t* peer_de, uint64_t* out);
int arkmpc_group_hostmul_abort(arkmpc_group_hostmul* session);
/* device <-> device over xGMI as DIRECT PEER WRITES (no ring): gather = every member pushes its range into the full buffer on
 * member `root`; allgather = every member pushes its range into every member's full buffer (G*(G-1) point-to-point copies, all
 * links busy at once); scatter = root pushes each member its range of a full buffer.  The consumers' streams wait on the device. */
int arkmpc_group_gather(arkmpc_group* grp, size_t n, size_t segs, size_t elem_words, const uint64_t* const* shards, int root_member,
                        uint64_t* out_on_root);
int arkmpc_group_allgather(arkmpc_group* grp, size_t n, size_t segs, size_t elem_words, const uint64_t* const* shards, uint64_t* const* outs);
int arkmpc_group_scatter(arkmpc_group* grp, size_t n, size_t segs, size_t elem_words, const uint64_t* src_on_root, int root_member,
                         uint64_t* const* shards);
/* Beaver multiplication, range-sharded (authenticated_scalar.rs:848-879): x, y, a, b, c, out = sharded ScalarShare vectors in
 * `layout`; my_de / peer_de / out_de = sharded d||e vectors.  In a two-party deployment the d||e payload leaves through
 * arkmpc_group_gather_d2h and the peer's arrives through arkmpc_group_scatter_h2d; two in-process parties built on the same devices
 * hand each other their shard pointers member by member (the device form of network/mock.rs) and order the hand-over with
 * arkmpc_group_wait_group -- each party's group has its own streams, nothing else orders one party's K1 before the other's K2+K3. */
int arkmpc_group_beaver_mask(arkmpc_group* grp, int layout, size_t n, const uint64_t* const* x, const uint64_t* const* y,
                             const uint64_t* const* a, const uint64_t* const* b, uint64_t* const* out_de);
/* K1 whose stores ARE the gather: every member's kernel writes its d / e range straight into the full 2n-Scalar buffer on `root`
 * through the peer mapping.  ARKMPC_ERR_UNSUPPORTED without peer access to root (use _mask + _gather). */
int arkmpc_group_beaver_mask_gathered(arkmpc_group* grp, int layout, size_t n, const uint64_t* const* x, const uint64_t* const* y,
                                      const uint64_t* const* a, const uint64_t* const* b, int root_member, uint64_t* out_de_on_root);
int arkmpc_group_beaver_finish_fused(arkmpc_group* grp, int layout, size_t n, int party_id, const uint64_t mac_key[4],
                                     const uint64_t* const* my_de, const uint64_t* const* peer_de, const uint64_t* const* a,
                                     const uint64_t* const* b, const uint64_t* const* c, uint64_t* const* out);
/* batch open + MAC check, range-sharded (authenticated_scalar.rs:278-354) */
int arkmpc_group_share_extract(arkmpc_group* grp, int layout, size_t n, const uint64_t* const* shares, uint64_t* const* out_values);
int arkmpc_group_open_and_mac_check(arkmpc_group* grp, int layout, size_t n, const uint64_t mac_key[4], const uint64_t* const* shares,
                                    const uint64_t* const* peer_values, uint64_t* const* out_opened, uint64_t* const* out_chk);
/* K5 on every range, one synchronisation per member, AND of the members' flags; blocks */
int arkmpc_group_mac_verify(arkmpc_group* grp, size_t n, const uint64_t* const* mine, const uint64_t* const* peer, int* out_ok);
/* H1 over a sharded Scalar vector: the sponge absorbs member 0's range, then member 1's, ... (= the full vector in index order);
 * each member converts and DMAs its own range to pinned host memory, so no device gather is needed; blocks */
int arkmpc_group_commit_sha3(arkmpc_group* grp, size_t n, const uint64_t* const* values, const uint64_t blinder[4],
                             uint64_t out_commitment[4]);
/* CurvePoint::msm over sharded (point, scalar) pairs (curve.rs:549-560): one bucket MSM per member, the G partial points added on
 * member 0; out_point = HOST pointer to 12 x u64; blocks.  BN254 Fr groups. */
int arkmpc_group_g1_msm(arkmpc_group* grp, size_t n, const uint64_t* const* points, const uint64_t* const* scalars, uint64_t out_point[12]);
/* the same on a CURVE25519_FR group: 16-word extended points (curve.rs:549-560 is generic over C) */
int arkmpc_group_ed_msm(arkmpc_group* grp, size_t n, const uint64_t* const* points, const uint64_t* const* scalars, uint64_t out_point[16]);

#ifdef __cplusplus
}
#endif
#endif /* ARKMPC_H */
