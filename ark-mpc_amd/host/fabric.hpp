// fabric.hpp -- C++ host-side mirror of the slice of ark-mpc's public API that sits on the hot path, written above
// the C ABI (include/arkmpc.h).  The reference is Rust and the image has no Rust toolchain, so this is the host
// language fallback: same names, argument meaning and error behaviour as the reference, batch-granular.
//
//   reference (online-phase/src/...)                              here
//   ------------------------------------------------------------  ------------------------------------------
//   MpcFabric<C>                       fabric.rs:164              arkmpc::MpcFabric
//   PreprocessingPhase<C>              offline_prep.rs:12-82      arkmpc::PreprocessingPhase
//   PartyIDBeaverSource                offline_prep.rs:88-170     arkmpc::PartyIDBeaverSource
//   MpcNetwork / UnboundedDuplexStream network.rs:148-157, network/mock.rs:63-88   arkmpc::MpcNetwork / MockNetwork
//   Vec<ScalarResult<C>>               scalar_result.rs           arkmpc::ScalarBatch         (device resident)
//   Vec<AuthenticatedScalarResult<C>>  authenticated_scalar.rs    arkmpc::AuthenticatedScalarBatch
//   MpcError::AuthenticationError      error.rs:8-18              arkmpc::MpcError
//   execute_mock_mpc                   lib.rs:116-128             arkmpc::execute_mock_mpc
//
// What is NOT mirrored (out of scope): the DAG executor and ResultHandle futures -- operations
// here run eagerly on the GPU stream, and a "network op" is a blocking send/receive of the batch payload.
#pragma once
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "arkmpc.h"

namespace arkmpc {

using PartyId = uint64_t;
constexpr PartyId PARTY0 = 0;  // lib.rs:41
constexpr PartyId PARTY1 = 1;  // lib.rs:43

enum class MpcError { None, NetworkError, AuthenticationError, ArithmeticError };  // error.rs:8-18

struct Scalar {  // scalar.rs:46 -- 4 x u64 Montgomery limbs
    uint64_t l[4];
};
struct ScalarShare {  // scalar/share.rs:32-37
    Scalar share, mac;
};
static_assert(sizeof(ScalarShare) == 64, "arkworks layout");

inline void check(arkmpc_ctx* ctx, int rc, const char* what) {
    if (rc != ARKMPC_OK) throw std::runtime_error(std::string(what) + ": arkmpc status " + std::to_string(rc) + " " + arkmpc_last_error(ctx));
}

// ---------------------------------------------------------------------------------------------------------------
// Engine handle: one device context (device pointers) + one staging context (host pointers) per party
// ---------------------------------------------------------------------------------------------------------------
class Engine {
  public:
    Engine(int field_id, int device) : field_id_(field_id) {
        int rc = arkmpc_ctx_create(field_id, device, &dev_);
        if (rc != ARKMPC_OK) throw std::runtime_error("arkmpc_ctx_create failed (" + std::to_string(rc) + "): the engine needs a GPU, there is no CPU fallback");
        rc = arkmpc_ctx_create(field_id, device, &host_);
        if (rc != ARKMPC_OK) throw std::runtime_error("arkmpc_ctx_create failed");
        check(host_, arkmpc_ctx_set_host_buffers(host_, 1), "set_host_buffers");
    }
    ~Engine() {
        if (dev_) arkmpc_ctx_destroy(dev_);
        if (host_) arkmpc_ctx_destroy(host_);
    }
    Engine(const Engine&) = delete;
    arkmpc_ctx* dev() const { return dev_; }
    arkmpc_ctx* host() const { return host_; }
    int field_id() const { return field_id_; }
    // Scalar::from(u64) and friends: canonical little-endian integers -> Montgomery form (scalar.rs:131-139)
    std::vector<Scalar> from_canonical(const std::vector<Scalar>& c) const {
        std::vector<Scalar> out(c.size());
        if (!c.empty()) check(host_, arkmpc_scalar_from_canonical(host_, c.size(), &c[0].l[0], &out[0].l[0]), "from_canonical");
        return out;
    }
    std::vector<Scalar> to_canonical(const std::vector<Scalar>& m) const {
        std::vector<Scalar> out(m.size());
        if (!m.empty()) check(host_, arkmpc_scalar_to_canonical(host_, m.size(), &m[0].l[0], &out[0].l[0]), "to_canonical");
        return out;
    }
    Scalar from_u64(uint64_t v) const { return from_canonical({Scalar{{v, 0, 0, 0}}})[0]; }

  private:
    int field_id_;
    arkmpc_ctx* dev_ = nullptr;
    arkmpc_ctx* host_ = nullptr;
};

// RAII owner of ONE device batch = the carrier handle of the C ABI (arkmpc_batch, what a ResultValue::DeviceBatch holds,
// fabric/result.rs:47-64): element kind + count + storage.  Typed batches (ScalarBatch, AuthenticatedScalarBatch, PointBatch ...)
// are created with their kind; engine scratch and byte payloads use ARKMPC_KIND_WORDS (n raw u64 words).
class DeviceBuf {
  public:
    DeviceBuf() = default;
    DeviceBuf(std::shared_ptr<Engine> e, size_t words) : DeviceBuf(std::move(e), ARKMPC_KIND_WORDS, words) {}
    DeviceBuf(std::shared_ptr<Engine> e, int kind, size_t n, int layout = ARKMPC_LAYOUT_AOS) : e_(std::move(e)) {
        check(e_->dev(), arkmpc_batch_create(e_->dev(), kind, layout, n, &b_), "batch_create");
        words_ = n * arkmpc_batch_elem_words(b_);
    }
    // a second owner of the same storage (Clone of the handle: arkmpc_batch_retain)
    DeviceBuf share() const { DeviceBuf d; if (b_) { arkmpc_batch_retain(b_); d.e_ = e_; d.b_ = b_; d.words_ = words_; } return d; }
    // take over a handle the C ABI returned (arkmpc_batch_from_host, arkmpc_batch_column, ...)
    static DeviceBuf adopt(std::shared_ptr<Engine> e, arkmpc_batch* b) {
        DeviceBuf d; d.e_ = std::move(e); d.b_ = b; d.words_ = arkmpc_batch_len(b) * arkmpc_batch_elem_words(b); return d;
    }
    ~DeviceBuf() { if (b_) arkmpc_batch_destroy(e_->dev(), b_); }
    DeviceBuf(DeviceBuf&& o) noexcept { *this = std::move(o); }
    DeviceBuf& operator=(DeviceBuf&& o) noexcept {
        if (this != &o) { if (b_) arkmpc_batch_destroy(e_->dev(), b_); e_ = std::move(o.e_); b_ = o.b_; words_ = o.words_; o.b_ = nullptr; o.words_ = 0; }
        return *this;
    }
    DeviceBuf(const DeviceBuf&) = delete;
    uint64_t* ptr() const { return arkmpc_batch_data(b_); }
    uint64_t* mac_ptr() const { return arkmpc_batch_mac_data(b_); }      // ScalarShare batches: the MAC half / column
    size_t stride() const { return arkmpc_batch_stride(b_); }           // u64 units between consecutive elements
    int layout() const { return arkmpc_batch_layout(b_); }
    arkmpc_batch* handle() const { return b_; }
    int kind() const { return arkmpc_batch_kind(b_); }
    size_t words() const { return words_; }
    void set_words(size_t w) { words_ = w; }          // logical length of a buffer allocated with slack
    // a batch handed over by the peer party: from now on this engine's stream uses it, so this engine drops it (the storage
    // returns to the pool ordered on the dropping context's stream)
    void rebind(std::shared_ptr<Engine> e) { e_ = std::move(e); }
    void upload(const void* host, size_t bytes) { if (bytes) check(e_->dev(), arkmpc_memcpy_h2d(e_->dev(), ptr(), host, bytes), "h2d"); }
    void download(void* host, size_t bytes) const {          // the copy runs on the context's stream, behind everything submitted so far, and blocks
        if (bytes) check(e_->dev(), arkmpc_memcpy_d2h(e_->dev(), host, ptr(), bytes), "d2h");
        else check(e_->dev(), arkmpc_sync(e_->dev()), "sync");
    }

  private:
    std::shared_ptr<Engine> e_;
    arkmpc_batch* b_ = nullptr;
    size_t words_ = 0;
};

// ---------------------------------------------------------------------------------------------------------------
// Network: network.rs:148-157.  Payloads on this path are batches of scalars (NetworkPayload::ScalarBatch, :45-60).
// ---------------------------------------------------------------------------------------------------------------
struct NetworkOutbound {
    uint64_t result_id;            // ids are allocated in lock-step by both parties (fabric.rs:282-295)
    std::vector<Scalar> payload;
    // wire mode (MpcFabric::set_wire_frames): the message as QuicTwoPartyNet would put it on the stream -- u64 LE length +
    // serde_json text (network/quic.rs:303-306), produced and parsed by the engine's codec (csrc/arkmpc_wire.hip)
    std::vector<uint8_t> frame;
    // device mode (LinkMode::Device): the batch stays in HBM and the message carries its buffer -- the in-memory move of
    // network/mock.rs:63-88 for a GPU-resident value (both parties of the mock run in one process on one GPU)
    std::shared_ptr<DeviceBuf> dev;
    // ... ordered after the sender's stream by an event the receiver's stream waits on (device side; neither host thread blocks)
    std::shared_ptr<arkmpc_event> ready;
};
class MpcNetwork {
  public:
    virtual ~MpcNetwork() = default;
    virtual PartyId party_id() const = 0;
    virtual void send(NetworkOutbound&& msg) = 0;
    virtual NetworkOutbound receive() = 0;     // blocks; throws on a closed peer (MpcNetworkError::RecvError)
    virtual void close() = 0;
};
// network/mock.rs:63-88: a pair of unbounded in-memory queues
struct DuplexQueue {
    std::mutex mu;
    std::condition_variable cv;
    std::deque<NetworkOutbound> q;
    bool closed = false;
    std::atomic<int> pending{0};           // mirrors q.size() / closed for the lock-free poll in receive()
    std::atomic<bool> closed_flag{false};
};
class MockNetwork : public MpcNetwork {
  public:
    MockNetwork(PartyId id, std::shared_ptr<DuplexQueue> out, std::shared_ptr<DuplexQueue> in) : id_(id), out_(std::move(out)), in_(std::move(in)) {}
    PartyId party_id() const override { return id_; }
    void send(NetworkOutbound&& msg) override {
        { std::lock_guard<std::mutex> lk(out_->mu); out_->q.push_back(std::move(msg)); out_->pending.fetch_add(1, std::memory_order_release); }
        out_->cv.notify_one();
    }
    NetworkOutbound receive() override {
        // a round trip of the latency-bound circuits (one network op per sequential gate) is ~100 us, of which a condition-
        // variable wake-up is 20-50: poll the queue briefly before blocking
        for (int spin = 0; spin < 20000; ++spin) {
            if (in_->pending.load(std::memory_order_acquire) > 0 || in_->closed_flag.load(std::memory_order_acquire)) break;
#if defined(__x86_64__)
            __builtin_ia32_pause();
#endif
        }
        std::unique_lock<std::mutex> lk(in_->mu);
        in_->cv.wait(lk, [&] { return !in_->q.empty() || in_->closed; });
        if (in_->q.empty()) throw std::runtime_error("MpcNetworkError::RecvError: peer closed");
        NetworkOutbound m = std::move(in_->q.front());
        in_->q.pop_front();
        in_->pending.fetch_sub(1, std::memory_order_release);
        return m;
    }
    void close() override {
        { std::lock_guard<std::mutex> lk(out_->mu); out_->closed = true; out_->closed_flag.store(true, std::memory_order_release); }
        out_->cv.notify_all();
    }

  private:
    PartyId id_;
    std::shared_ptr<DuplexQueue> out_, in_;
};

// ---------------------------------------------------------------------------------------------------------------
// Preprocessing: offline_prep.rs:12-82
// ---------------------------------------------------------------------------------------------------------------
// thrown by a source that cannot serve a request (LowGearPrep asserts, offline-phase/src/structs.rs:189); sources throw it BEFORE consuming anything
struct PreprocessingExhausted : std::runtime_error {
    using std::runtime_error::runtime_error;
};
class PreprocessingPhase {
  public:
    virtual ~PreprocessingPhase() = default;
    virtual Scalar get_mac_key_share() = 0;
    virtual std::pair<std::vector<Scalar>, std::vector<ScalarShare>> next_local_input_mask_batch(size_t n) = 0;
    virtual std::vector<ScalarShare> next_counterparty_input_mask_batch(size_t n) = 0;
    virtual void next_triplet_batch(size_t n, std::vector<ScalarShare>& a, std::vector<ScalarShare>& b, std::vector<ScalarShare>& c) = 0;
    virtual std::vector<ScalarShare> next_shared_bit_batch(size_t n) = 0;                         // offline_prep.rs:39-44
    virtual std::vector<ScalarShare> next_shared_value_batch(size_t n) = 0;                       // offline_prep.rs:45-50
    virtual void next_shared_inverse_pair_batch(size_t n, std::vector<ScalarShare>& l, std::vector<ScalarShare>& r) = 0;   // :54-60
    // Optional: a source whose batches are n copies of one value (the dummy source below) may describe them by that value;
    // the fabric then fills the batch on the GPU (arkmpc_fill) instead of building and uploading n-element host vectors.
    virtual bool constant_triplet(ScalarShare&, ScalarShare&, ScalarShare&) { return false; }
    // Optional: a source that keeps its triples in memory of its own (LowGearPrep holds Vecs it split_off()s from, offline-phase
    // lowgear/mod.rs) may LEND the next n instead of copying them out: the three pointers address n consecutive ScalarShare records each, valid
    // and unmodified until the next call on this source.  The fabric then imports them straight from there (asynchronously, in place over
    // the link when the storage is pinned) -- no intermediate Vec, no pin / unpin per gate.  Consumes the triples like next_triplet_batch.
    virtual bool borrow_triplet_batch(size_t /*n*/, const ScalarShare** /*a*/, const ScalarShare** /*b*/, const ScalarShare** /*c*/) { return false; }
    // Optional: how many triples the source can still hand out (SIZE_MAX = it does not say).  The fabric reads triples AHEAD of need only when the
    // source can serve them: a read-ahead that would exhaust the source is not attempted, so nothing is consumed for a gate that may never come.
    virtual size_t triples_remaining() const { return SIZE_MAX; }
    virtual bool constant_input_masks(Scalar& /*local value*/, ScalarShare& /*local share*/, ScalarShare& /*counterparty share*/) { return false; }
};
// offline_prep.rs:88-170: a = 2, b = 3, c = 6 statically split; MAC key share = party id
class PartyIDBeaverSource : public PreprocessingPhase {
  public:
    PartyIDBeaverSource(PartyId party, const Engine& e) : party_(party) {
        for (uint64_t v = 0; v <= 6; ++v) s_[v] = e.from_u64(v);
    }
    Scalar get_mac_key_share() override { return s_[party_]; }                                   // :108-110
    std::pair<std::vector<Scalar>, std::vector<ScalarShare>> next_local_input_mask_batch(size_t n) override {  // :112-119
        const Scalar v = s_[3], pv = s_[3 * party_];
        return {std::vector<Scalar>(n, v), std::vector<ScalarShare>(n, ScalarShare{pv, pv})};
    }
    std::vector<ScalarShare> next_counterparty_input_mask_batch(size_t n) override {            // :121-127
        const Scalar pv = s_[3 * party_];                                                        // value = 3*party, mac = party*value
        return std::vector<ScalarShare>(n, ScalarShare{pv, pv});
    }
    void next_triplet_batch(size_t n, std::vector<ScalarShare>& a, std::vector<ScalarShare>& b, std::vector<ScalarShare>& c) override {  // :137-158
        const uint64_t k = party_;   // key share
        ScalarShare ta, tb, tc;
        if (party_ == 0) { ta.share = s_[1]; tb.share = s_[3]; tc.share = s_[2]; }
        else { ta.share = s_[1]; tb.share = s_[0]; tc.share = s_[4]; }
        ta.mac = s_[k * 2]; tb.mac = s_[k * 3]; tc.mac = s_[k * 6];
        a.assign(n, ta); b.assign(n, tb); c.assign(n, tc);
    }

    bool constant_triplet(ScalarShare& a, ScalarShare& b, ScalarShare& c) override {
        std::vector<ScalarShare> va, vb, vc;
        next_triplet_batch(1, va, vb, vc);
        a = va[0]; b = vb[0]; c = vc[0];
        return true;
    }
    bool constant_input_masks(Scalar& v, ScalarShare& local, ScalarShare& counterparty) override {
        auto lm = next_local_input_mask_batch(1);
        v = lm.first[0]; local = lm.second[0]; counterparty = next_counterparty_input_mask_batch(1)[0];
        return true;
    }

    std::vector<ScalarShare> next_shared_bit_batch(size_t n) override {                         // :131-135: "simply output partyID"
        return std::vector<ScalarShare>(n, ScalarShare{s_[party_], s_[party_]});
    }
    std::vector<ScalarShare> next_shared_value_batch(size_t n) override {                       // :166-168: (party_id, party_id)
        return std::vector<ScalarShare>(n, ScalarShare{s_[party_], s_[party_]});
    }
    void next_shared_inverse_pair_batch(size_t n, std::vector<ScalarShare>& l, std::vector<ScalarShare>& r) override {   // :159-164: 1 * 1 = 1
        l.assign(n, ScalarShare{s_[party_], s_[party_]}); r.assign(n, ScalarShare{s_[party_], s_[party_]});
    }

  private:
    PartyId party_;
    Scalar s_[7];
};

// A trusted-dealer source for tests and benches with NON-degenerate material (the dummy source above has MAC key shares 0 / 1 and
// constant triples, so party 0's MACs are all zero): both parties construct it with the same seed, derive the same random values
// in the same order -- MAC key shares, triples c = a * b, masks, bits, inverse pairs -- with random additive splits of every value
// and of mac_key * value, and each keeps its own halves.  What PreprocessingPhase promises (offline_prep.rs:12-82) and nothing more;
// the arithmetic runs on the engine's host-pointer context.
class DealerBeaverSource : public PreprocessingPhase {
  public:
    DealerBeaverSource(PartyId party, const Engine& e, uint64_t seed) : party_(party), e_(e), state_(seed ^ 0xD1B54A32D192ED03ull) {
        std::vector<Scalar> k = random(2);
        key_share_[0] = k[0]; key_share_[1] = k[1];
        key_ = add(std::vector<Scalar>{k[0]}, std::vector<Scalar>{k[1]})[0];
    }
    Scalar get_mac_key_share() override { return key_share_[party_]; }
    std::pair<std::vector<Scalar>, std::vector<ScalarShare>> next_local_input_mask_batch(size_t n) override {
        std::vector<Scalar> v = random(n);
        return {v, split(v)};
    }
    std::vector<ScalarShare> next_counterparty_input_mask_batch(size_t n) override { return split(random(n)); }
    void next_triplet_batch(size_t n, std::vector<ScalarShare>& a, std::vector<ScalarShare>& b, std::vector<ScalarShare>& c) override {
        std::vector<Scalar> va = random(n), vb = random(n);
        a = split(va); b = split(vb); c = split(mul(va, vb));
    }
    std::vector<ScalarShare> next_shared_bit_batch(size_t n) override {
        std::vector<Scalar> bits(n);
        for (size_t i = 0; i < n; ++i) bits[i] = Scalar{{next() & 1u, 0, 0, 0}};
        return split(e_.from_canonical(bits));
    }
    std::vector<ScalarShare> next_shared_value_batch(size_t n) override { return split(random(n)); }
    void next_shared_inverse_pair_batch(size_t n, std::vector<ScalarShare>& l, std::vector<ScalarShare>& r) override {
        std::vector<Scalar> v = random(n), inv(n);                    // a random element is zero with probability ~2^-252
        if (n) check(e_.host(), arkmpc_scalar_batch_inverse(e_.host(), n, &v[0].l[0], &inv[0].l[0]), "batch_inverse");
        l = split(v); r = split(inv);
    }

  private:
    uint64_t next() {                                                 // splitmix64
        uint64_t z = (state_ += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    std::vector<Scalar> random(size_t n) {                            // uniform below 2^252 (< every modulus of the engine), Montgomery form
        std::vector<Scalar> c(n);
        for (size_t i = 0; i < n; ++i) c[i] = Scalar{{next(), next(), next(), next() >> 12}};
        return e_.from_canonical(c);
    }
    typedef int (*BinOp)(arkmpc_ctx*, size_t, const uint64_t*, const uint64_t*, uint64_t*);
    std::vector<Scalar> bin(BinOp f, const std::vector<Scalar>& a, const std::vector<Scalar>& b) const {
        std::vector<Scalar> o(a.size());
        if (!a.empty()) check(e_.host(), f(e_.host(), a.size(), &a[0].l[0], &b[0].l[0], &o[0].l[0]), "dealer arithmetic");
        return o;
    }
    std::vector<Scalar> add(const std::vector<Scalar>& a, const std::vector<Scalar>& b) const { return bin(arkmpc_scalar_add, a, b); }
    std::vector<Scalar> mul(const std::vector<Scalar>& a, const std::vector<Scalar>& b) const { return bin(arkmpc_scalar_mul, a, b); }
    // this party's (share, mac) halves of additive splits of v and of mac_key * v
    std::vector<ScalarShare> split(const std::vector<Scalar>& v) {
        const size_t n = v.size();
        std::vector<Scalar> r = random(n), t = random(n), macs = mul(std::vector<Scalar>(n, key_), v);
        std::vector<ScalarShare> out(n);
        if (party_ == 0) { for (size_t i = 0; i < n; ++i) out[i] = ScalarShare{r[i], t[i]}; return out; }
        std::vector<Scalar> s1 = bin(arkmpc_scalar_sub, v, r), m1 = bin(arkmpc_scalar_sub, macs, t);
        for (size_t i = 0; i < n; ++i) out[i] = ScalarShare{s1[i], m1[i]};
        return out;
    }
    PartyId party_;
    const Engine& e_;
    uint64_t state_;
    Scalar key_share_[2], key_;
};

// A source that holds its triples in PINNED host memory, the way a real offline phase would leave them for this engine: `capacity` triples are
// drawn from an inner source once (here: at construction) into three arkmpc_host_alloc blocks and then handed out in order -- copied into Vecs
// through the PreprocessingPhase interface (next_triplet_batch), or lent in place (borrow_triplet_batch).  Everything else -- MAC key share,
// masks, bits, inverse pairs -- is the inner source's.  Running out is what LowGearPrep does (assert, offline-phase/src/structs.rs:189): throws.
class VectorBeaverSource : public PreprocessingPhase {
  public:
    VectorBeaverSource(std::unique_ptr<PreprocessingPhase> inner, size_t capacity) : inner_(std::move(inner)), cap_(capacity) {
        std::vector<ScalarShare> a, b, c;
        inner_->next_triplet_batch(cap_, a, b, c);
        if (a.size() != cap_ || b.size() != cap_ || c.size() != cap_) throw std::runtime_error("VectorBeaverSource: the inner source is short of triples");
        const std::vector<ScalarShare>* src[3] = {&a, &b, &c};
        for (int k = 0; k < 3; ++k) {
            void* q = nullptr;
            if (arkmpc_host_alloc((cap_ ? cap_ : 1) * sizeof(ScalarShare), &q) != ARKMPC_OK) throw std::runtime_error("arkmpc_host_alloc failed: no GPU runtime, or out of pinnable memory");
            store_[k] = static_cast<ScalarShare*>(q);
            if (cap_) std::memcpy(store_[k], src[k]->data(), cap_ * sizeof(ScalarShare));
        }
    }
    ~VectorBeaverSource() override { for (auto* q : store_) if (q) arkmpc_host_free(q); }
    VectorBeaverSource(const VectorBeaverSource&) = delete;
    Scalar get_mac_key_share() override { return inner_->get_mac_key_share(); }
    std::pair<std::vector<Scalar>, std::vector<ScalarShare>> next_local_input_mask_batch(size_t n) override { return inner_->next_local_input_mask_batch(n); }
    std::vector<ScalarShare> next_counterparty_input_mask_batch(size_t n) override { return inner_->next_counterparty_input_mask_batch(n); }
    std::vector<ScalarShare> next_shared_bit_batch(size_t n) override { return inner_->next_shared_bit_batch(n); }
    std::vector<ScalarShare> next_shared_value_batch(size_t n) override { return inner_->next_shared_value_batch(n); }
    void next_shared_inverse_pair_batch(size_t n, std::vector<ScalarShare>& l, std::vector<ScalarShare>& r) override { inner_->next_shared_inverse_pair_batch(n, l, r); }
    void next_triplet_batch(size_t n, std::vector<ScalarShare>& a, std::vector<ScalarShare>& b, std::vector<ScalarShare>& c) override {
        const ScalarShare *pa, *pb, *pc;
        borrow_triplet_batch(n, &pa, &pb, &pc);
        a.assign(pa, pa + n); b.assign(pb, pb + n); c.assign(pc, pc + n);
    }
    bool borrow_triplet_batch(size_t n, const ScalarShare** a, const ScalarShare** b, const ScalarShare** c) override {
        if (n > cap_ - used_) throw PreprocessingExhausted("preprocessing exhausted: " + std::to_string(cap_ - used_) + " triples left, " + std::to_string(n) + " requested");
        *a = store_[0] + used_; *b = store_[1] + used_; *c = store_[2] + used_;
        used_ += n;
        return true;
    }
    size_t remaining() const { return cap_ - used_; }
    size_t triples_remaining() const override { return cap_ - used_; }

  private:
    std::unique_ptr<PreprocessingPhase> inner_;
    size_t cap_ = 0, used_ = 0;
    ScalarShare* store_[3] = {nullptr, nullptr, nullptr};
};

// ---------------------------------------------------------------------------------------------------------------
// Curve traits: the reference is generic over `C: CurveGroup` (authenticated_curve.rs, curve.rs); the C ABI has one set of
// entry points per curve.  A traits struct binds the point-side API of the mirror to one of them.
// ---------------------------------------------------------------------------------------------------------------
struct Bn254G1 {                      // ark_bn254::G1Projective: short-Weierstrass Jacobian {x, y, z}
    static constexpr size_t PW = 12;  // u64 words per CurvePoint
    static constexpr int FIELD = ARKMPC_BN254_FR;
    static int add(arkmpc_ctx* c, size_t n, const uint64_t* a, const uint64_t* b, uint64_t* o) { return arkmpc_g1_add(c, n, a, b, o); }
    static int sub(arkmpc_ctx* c, size_t n, const uint64_t* a, const uint64_t* b, uint64_t* o) { return arkmpc_g1_sub(c, n, a, b, o); }
    static int scalar_mul(arkmpc_ctx* c, size_t n, const uint64_t* p, const uint64_t* s, uint64_t* o) { return arkmpc_g1_scalar_mul(c, n, p, s, o); }
    static int generator_mul(arkmpc_ctx* c, size_t n, const uint64_t* s, uint64_t* o) { return arkmpc_g1_generator_mul(c, n, s, o); }
    static int to_bytes(arkmpc_ctx* c, size_t n, const uint64_t* p, uint8_t* o) { return arkmpc_g1_to_bytes(c, n, p, o); }
    static int from_bytes(arkmpc_ctx* c, size_t n, const uint8_t* b, uint64_t* o, uint8_t* ok) { return arkmpc_g1_from_bytes(c, n, b, o, ok); }
    static int share_add(arkmpc_ctx* c, size_t n, const uint64_t* a, const uint64_t* b, uint64_t* o) { return arkmpc_pointshare_add(c, n, a, b, o); }
    static int share_sub(arkmpc_ctx* c, size_t n, const uint64_t* a, const uint64_t* b, uint64_t* o) { return arkmpc_pointshare_sub(c, n, a, b, o); }
    static int share_neg(arkmpc_ctx* c, size_t n, const uint64_t* a, uint64_t* o) { return arkmpc_pointshare_neg(c, n, a, o); }
    static int share_add_public(arkmpc_ctx* c, size_t n, int party, const uint64_t* k, const uint64_t* a, const uint64_t* p, uint64_t* o) { return arkmpc_pointshare_add_public(c, n, party, k, a, p, o); }
    static int share_sub_public(arkmpc_ctx* c, size_t n, int party, const uint64_t* k, const uint64_t* a, const uint64_t* p, uint64_t* o) { return arkmpc_pointshare_sub_public(c, n, party, k, a, p, o); }
    static int share_mul_public(arkmpc_ctx* c, size_t n, const uint64_t* a, const uint64_t* s, uint64_t* o) { return arkmpc_pointshare_mul_public(c, n, a, s, o); }
    static int scalarshare_mul_generator(arkmpc_ctx* c, size_t n, const uint64_t* ss, uint64_t* o) { return arkmpc_scalarshare_mul_generator(c, n, ss, o); }
    static int scalarshare_mul_point(arkmpc_ctx* c, size_t n, const uint64_t* ss, const uint64_t* p, uint64_t* o) { return arkmpc_scalarshare_mul_point(c, n, ss, p, o); }
    static int share_extract(arkmpc_ctx* c, size_t n, const uint64_t* a, uint64_t* o) { return arkmpc_pointshare_extract(c, n, a, o); }
    static int mac_check_shares(arkmpc_ctx* c, size_t n, const uint64_t* k, const uint64_t* v, const uint64_t* a, uint64_t* o) { return arkmpc_point_mac_check_shares(c, n, k, v, a, o); }
    static int commit_points(arkmpc_ctx* c, size_t n, const uint64_t* p, const uint64_t* bl, uint64_t* o) { return arkmpc_commit_points_sha3(c, n, p, bl, o); }
    static int mac_verify(arkmpc_ctx* c, size_t n, const uint64_t* a, const uint64_t* b, uint8_t* ok) { return arkmpc_point_mac_verify(c, n, a, b, ok); }
    static int share_sum(arkmpc_ctx* c, size_t n, const uint64_t* a, uint64_t* o) { return arkmpc_pointshare_sum(c, n, a, o); }
    static int beaver_finish(arkmpc_ctx* c, size_t n, int party, const uint64_t* k, const uint64_t* d, const uint64_t* eG, const uint64_t* ta, const uint64_t* tb,
                             const uint64_t* tc, uint64_t* o) { return arkmpc_point_beaver_finish(c, n, party, k, d, eG, ta, tb, tc, o); }
    // CurvePoint::msm / msm_authenticated (curve.rs:549-560, :618-642): the bucket method on the GPU
    static int msm(arkmpc_ctx* c, size_t n, const uint64_t* p, const uint64_t* s, uint64_t* o) { return arkmpc_g1_msm(c, n, p, s, o); }
    static int msm_authenticated(arkmpc_ctx* c, size_t n, const uint64_t* p, const uint64_t* ss, uint64_t* o) { return arkmpc_g1_msm_authenticated(c, n, p, ss, o); }
};
struct Curve25519 {                   // ark_curve25519::EdwardsProjective (README.md:24): extended twisted Edwards {x, y, t, z}
    static constexpr size_t PW = 16;
    static constexpr int FIELD = ARKMPC_CURVE25519_FR;
    static int add(arkmpc_ctx* c, size_t n, const uint64_t* a, const uint64_t* b, uint64_t* o) { return arkmpc_ed_add(c, n, a, b, o); }
    static int sub(arkmpc_ctx* c, size_t n, const uint64_t* a, const uint64_t* b, uint64_t* o) { return arkmpc_ed_sub(c, n, a, b, o); }
    static int scalar_mul(arkmpc_ctx* c, size_t n, const uint64_t* p, const uint64_t* s, uint64_t* o) { return arkmpc_ed_scalar_mul(c, n, p, s, o); }
    static int generator_mul(arkmpc_ctx* c, size_t n, const uint64_t* s, uint64_t* o) { return arkmpc_ed_generator_mul(c, n, s, o); }
    static int to_bytes(arkmpc_ctx* c, size_t n, const uint64_t* p, uint8_t* o) { return arkmpc_ed_to_bytes(c, n, p, o); }
    static int from_bytes(arkmpc_ctx* c, size_t n, const uint8_t* b, uint64_t* o, uint8_t* ok) { return arkmpc_ed_from_bytes(c, n, b, o, ok); }
    static int share_add(arkmpc_ctx* c, size_t n, const uint64_t* a, const uint64_t* b, uint64_t* o) { return arkmpc_edshare_add(c, n, a, b, o); }
    static int share_sub(arkmpc_ctx* c, size_t n, const uint64_t* a, const uint64_t* b, uint64_t* o) { return arkmpc_edshare_sub(c, n, a, b, o); }
    static int share_neg(arkmpc_ctx* c, size_t n, const uint64_t* a, uint64_t* o) { return arkmpc_edshare_neg(c, n, a, o); }
    static int share_add_public(arkmpc_ctx* c, size_t n, int party, const uint64_t* k, const uint64_t* a, const uint64_t* p, uint64_t* o) { return arkmpc_edshare_add_public(c, n, party, k, a, p, o); }
    static int share_sub_public(arkmpc_ctx* c, size_t n, int party, const uint64_t* k, const uint64_t* a, const uint64_t* p, uint64_t* o) { return arkmpc_edshare_sub_public(c, n, party, k, a, p, o); }
    static int share_mul_public(arkmpc_ctx* c, size_t n, const uint64_t* a, const uint64_t* s, uint64_t* o) { return arkmpc_edshare_mul_public(c, n, a, s, o); }
    static int scalarshare_mul_generator(arkmpc_ctx* c, size_t n, const uint64_t* ss, uint64_t* o) { return arkmpc_scalarshare_mul_ed_generator(c, n, ss, o); }
    static int scalarshare_mul_point(arkmpc_ctx* c, size_t n, const uint64_t* ss, const uint64_t* p, uint64_t* o) { return arkmpc_scalarshare_mul_ed_point(c, n, ss, p, o); }
    static int share_extract(arkmpc_ctx* c, size_t n, const uint64_t* a, uint64_t* o) { return arkmpc_edshare_extract(c, n, a, o); }
    static int mac_check_shares(arkmpc_ctx* c, size_t n, const uint64_t* k, const uint64_t* v, const uint64_t* a, uint64_t* o) { return arkmpc_ed_mac_check_shares(c, n, k, v, a, o); }
    static int commit_points(arkmpc_ctx* c, size_t n, const uint64_t* p, const uint64_t* bl, uint64_t* o) { return arkmpc_commit_ed_points_sha3(c, n, p, bl, o); }
    static int mac_verify(arkmpc_ctx* c, size_t n, const uint64_t* a, const uint64_t* b, uint8_t* ok) { return arkmpc_ed_mac_verify(c, n, a, b, ok); }
    static int share_sum(arkmpc_ctx* c, size_t n, const uint64_t* a, uint64_t* o) { return arkmpc_edshare_sum(c, n, a, o); }
    static int beaver_finish(arkmpc_ctx* c, size_t n, int party, const uint64_t* k, const uint64_t* d, const uint64_t* eG, const uint64_t* ta, const uint64_t* tb,
                             const uint64_t* tc, uint64_t* o) { return arkmpc_edpoint_beaver_finish(c, n, party, k, d, eG, ta, tb, tc, o); }
    // CurvePoint::msm / msm_authenticated (curve.rs:549-560, :618-642): the bucket method on the complete Edwards addition
    static int msm(arkmpc_ctx* c, size_t n, const uint64_t* p, const uint64_t* s, uint64_t* o) { return arkmpc_ed_msm(c, n, p, s, o); }
    static int msm_authenticated(arkmpc_ctx* c, size_t n, const uint64_t* p, const uint64_t* ss, uint64_t* o) { return arkmpc_ed_msm_authenticated(c, n, p, ss, o); }
};

// ---------------------------------------------------------------------------------------------------------------
// Fabric + result batches
// ---------------------------------------------------------------------------------------------------------------
class MpcFabric;

// Vec<ScalarResult<C>>: n public scalars resident on the GPU
struct ScalarBatch {
    size_t n = 0;
    DeviceBuf buf;
    std::vector<Scalar> to_host() const { std::vector<Scalar> v(n); buf.download(v.data(), n * 32); return v; }
};

// Sum / Product for ScalarResult (scalar_result.rs:325-338 and Iterator::sum): one gate over n public scalars -> a batch of ONE
inline ScalarBatch scalar_batch_reduce(const std::shared_ptr<Engine>& e, const ScalarBatch& v, bool product) {
    if (product && v.n == 0) throw std::invalid_argument("Cannot compute product of empty iterator");     // assert! scalar_result.rs:328
    ScalarBatch r; r.n = 1; r.buf = DeviceBuf(e, ARKMPC_KIND_SCALAR, 1);
    check(e->dev(), (product ? arkmpc_scalar_product : arkmpc_scalar_sum)(e->dev(), v.n, v.buf.ptr(), r.buf.ptr()), product ? "scalar_product" : "scalar_sum");
    return r;
}
// ScalarResult::batch_add_constant / batch_sub_constant (scalar_result.rs:119-137, :205-223): public batch (+|-) plain Scalars
inline ScalarBatch scalar_batch_addsub(const std::shared_ptr<Engine>& e, const ScalarBatch& a, const ScalarBatch& b, bool sub) {
    if (a.n != b.n) throw std::invalid_argument("Batch add constant requires equal length inputs");
    ScalarBatch r; r.n = a.n; r.buf = DeviceBuf(e, ARKMPC_KIND_SCALAR, a.n);
    if (a.n) check(e->dev(), (sub ? arkmpc_scalar_sub : arkmpc_scalar_add)(e->dev(), a.n, a.buf.ptr(), b.buf.ptr(), r.buf.ptr()), "scalar add/sub constant");
    return r;
}

class AuthenticatedScalarBatch;
struct AuthenticatedOpenResult {   // AuthenticatedScalarOpenResult, authenticated_scalar.rs:360-385
    MpcError err = MpcError::None; // AuthenticationError unless the MAC check scalar equals 1
    ScalarBatch value;
};

class MpcFabric : public std::enable_shared_from_this<MpcFabric> {
  public:
    MpcFabric(PartyId party, std::shared_ptr<Engine> eng, std::unique_ptr<MpcNetwork> net, std::unique_ptr<PreprocessingPhase> prep)
        : party_(party), eng_(std::move(eng)), net_(std::move(net)), prep_(std::move(prep)) {
        mac_key_ = prep_->get_mac_key_share();                                                   // fabric.rs:446
    }
    PartyId party_id() const { return party_; }
    const Scalar& mac_key() const { return mac_key_; }
    std::shared_ptr<Engine> engine() const { return eng_; }
    arkmpc_ctx* ctx() const { return eng_->dev(); }

    ScalarBatch allocate_scalars(const std::vector<Scalar>& mont) {                               // fabric.rs:652-668
        ScalarBatch b; b.n = mont.size(); b.buf = DeviceBuf(eng_, ARKMPC_KIND_SCALAR, mont.size()); b.buf.upload(mont.data(), mont.size() * 32);
        return b;
    }
    // Wire mode: every batch crosses the mock link as the serde_json frame the QUIC transport would carry, so the
    // protocol-level scenarios exercise the GPU encoder / validating decoder end to end (env ARKMPC_MOCK_WIRE=1 in
    // execute_mock_mpc).  Off: payloads are handed over as host vectors (network/mock.rs).
    enum class LinkMode { Host, Device, Wire };
    // HBM layout of every AuthenticatedScalarBatch this fabric creates: the engine-native split columns (default: K1 reads no dead MAC
    // bytes, the `.share()` payload of an opening IS the share column, once-streamed data carries cache hints) or arkworks' AoS records.
    // Values cross to the host (to_host, the mock link's host mode) as arkworks records either way.
    void set_share_layout(int layout) { share_layout_ = layout; }
    int share_layout() const { return share_layout_; }
    void set_link_mode(LinkMode m) { link_ = m; wire_ = (m == LinkMode::Wire); }
    LinkMode link_mode() const { return link_; }
    void set_wire_frames(bool on) { set_link_mode(on ? LinkMode::Wire : LinkMode::Host); }
    bool wire_frames() const { return wire_; }
    static std::atomic<uint64_t>& frames_sent() { static std::atomic<uint64_t> c{0}; return c; }   // process-wide, for the tests
    // send / receive / exchange of a scalar batch (fabric.rs:720-814): party 0 sends first then receives
    void send_values(const ScalarBatch& v) {
        if (link_ == LinkMode::Device) { send_device(v.buf, 4 * v.n); return; }
        if (!wire_) { net_->send(NetworkOutbound{next_id_++, v.to_host(), {}, {}}); return; }
        const uint64_t id = next_id_++;
        size_t cap = 0, len = 0;
        arkmpc_wire_frame_bound(v.n, &cap);
        DeviceBuf fr(eng_, cap / 8 + 1);
        check(ctx(), arkmpc_wire_encode_scalar_batch(ctx(), id, v.n, v.buf.ptr(), reinterpret_cast<uint8_t*>(fr.ptr()), cap, &len), "wire_encode_scalar_batch");
        NetworkOutbound m{id, {}, std::vector<uint8_t>(len), {}};
        fr.download(m.frame.data(), len);
        frames_sent()++;
        net_->send(std::move(m));
    }
    // device handoff: a private copy of the words (the sender keeps using its buffer); the receiver's stream waits for the copy
    // through an event (arkmpc_event_record / _wait) -- the sending thread does not block on its stream, which is what bounds the
    // round time of latency-bound circuits (one network round per sequential gate)
    void send_device(const DeviceBuf& src, size_t words) {
        auto copy = std::make_shared<DeviceBuf>(eng_, words ? words : 1);
        copy->set_words(words);
        if (words) check(ctx(), arkmpc_memcpy_d2d(ctx(), copy->ptr(), src.ptr(), words * 8), "d2d");
        arkmpc_event* ev = nullptr;
        check(ctx(), arkmpc_event_record(ctx(), &ev), "event_record");
        NetworkOutbound m{next_id_++, {}, {}, std::move(copy), std::shared_ptr<arkmpc_event>(ev, [](arkmpc_event* e) { arkmpc_event_destroy(e); })};
        net_->send(std::move(m));
    }
    // device hand-over of a buffer the sender GIVES UP (no copy): the message owns it from here on
    void send_device_owned(std::shared_ptr<DeviceBuf> buf, size_t words) {
        buf->set_words(words);
        arkmpc_event* ev = nullptr;
        check(ctx(), arkmpc_event_record(ctx(), &ev), "event_record");
        NetworkOutbound m{next_id_++, {}, {}, std::move(buf), std::shared_ptr<arkmpc_event>(ev, [](arkmpc_event* e) { arkmpc_event_destroy(e); })};
        net_->send(std::move(m));
    }
    void await_device(const NetworkOutbound& m) {
        if (m.ready) check(ctx(), arkmpc_event_wait(ctx(), m.ready.get()), "event_wait");
    }
    // `expect` = the element count the protocol step requires.  What arrives is the PEER's choice (a short, empty or oversized
    // payload would otherwise be read out of bounds by the kernels launched with the local n), so a mismatch is a network
    // error here, before any kernel sees the buffer (the reference's `.into()` casts panic on a malformed value,
    // fabric/result.rs:127-233).  kAnyCount: the caller checks.
    static constexpr size_t kAnyCount = ~(size_t)0;
    static constexpr uint64_t kNextId = ~(uint64_t)0;
    ScalarBatch receive_values(size_t expect = kAnyCount, uint64_t id = kNextId) {
        ScalarBatch b = receive_values_unchecked(id);
        if (expect != kAnyCount && b.n != expect)
            throw std::runtime_error("MpcNetworkError: peer sent " + std::to_string(b.n) + " scalars where " + std::to_string(expect) + " were expected");
        return b;
    }
    ScalarBatch receive_values_unchecked(uint64_t id = kNextId) {
        if (id == kNextId) id = next_id_++;
        NetworkOutbound m = net_->receive();
        if (link_ == LinkMode::Device) {
            if (!m.dev) throw std::runtime_error("MpcNetworkError: device link message without a buffer");
            await_device(m);
            ScalarBatch b; b.n = m.dev->words() / 4; b.buf = std::move(*m.dev); b.buf.rebind(eng_); return b;
        }
        if (!wire_) return allocate_scalars(m.payload);
        // the element count is only known after parsing: a scalar's text is at least 66 bytes ("[0,0,...,0]," )
        const size_t max_n = m.frame.size() / 66 + 1;
        DeviceBuf fr(eng_, m.frame.size() / 8 + 2);
        fr.upload(m.frame.data(), m.frame.size());
        ScalarBatch b; b.buf = DeviceBuf(eng_, ARKMPC_KIND_SCALAR, max_n);
        size_t n = 0; uint64_t rid = 0;
        check(ctx(), arkmpc_wire_decode_scalar_batch(ctx(), reinterpret_cast<const uint8_t*>(fr.ptr()), m.frame.size(), max_n, b.buf.ptr(), &n, &rid),
              "wire_decode_scalar_batch");
        if (rid != id) throw std::runtime_error("MpcNetworkError: result id mismatch on a received frame");
        b.n = n;
        return b;
    }
    // both parties send a batch of the same length (every exchange on this path is symmetric): the peer's must match
    // Result ids: party 0 allocates send-then-receive, party 1 receive-then-send (fabric.rs:752-767), so the two sides stay in lock-step.
    // In the reference both operations only enqueue; here receive blocks, so party 1 takes its receive id first, SENDS, and only then
    // waits -- neither party's message is held back by the other's (one network latency per round, not two).
    ScalarBatch exchange_values(const ScalarBatch& mine) {
        if (party_ == PARTY0) { send_values(mine); return receive_values(mine.n); }
        const uint64_t rid = next_id_++;
        send_values(mine);
        return receive_values(mine.n, rid);
    }
    // exchange_values for a device-link payload the caller gives up (see send_device_owned)
    ScalarBatch exchange_device_owned(std::shared_ptr<DeviceBuf> mine, size_t words, size_t expect_n) {
        if (party_ == PARTY0) { send_device_owned(std::move(mine), words); return receive_values(expect_n); }
        const uint64_t rid = next_id_++;
        send_device_owned(std::move(mine), words);
        return receive_values(expect_n, rid);
    }
    // a payload that is already a host vector (the group fabric gathers its ranges to the host itself): host link semantics in every mode
    std::vector<Scalar> exchange_host_values(std::vector<Scalar>&& mine) {
        const size_t expect = mine.size();
        uint64_t sid, rid;
        if (party_ == PARTY0) { sid = next_id_++; rid = next_id_++; } else { rid = next_id_++; sid = next_id_++; }
        (void)rid;
        net_->send(NetworkOutbound{sid, std::move(mine), {}, {}, {}});
        NetworkOutbound m = net_->receive();
        if (m.payload.size() != expect)
            throw std::runtime_error("MpcNetworkError: peer sent " + std::to_string(m.payload.size()) + " scalars where " + std::to_string(expect) + " were expected");
        return std::move(m.payload);
    }
    // the next n triples as host vectors (PreprocessingPhase::next_triplet_batch, offline_prep.rs:65-81), ids advanced as next_triple_batch
    PreprocessingPhase& preprocessing() { return *prep_; }
    void advance_ids(uint64_t count) { next_id_ += count; }          // result ids of values a caller took from the source itself (fabric.rs:894-915 allocates 3n per triple batch)
    void next_triple_host(size_t n, std::vector<ScalarShare>& a, std::vector<ScalarShare>& b, std::vector<ScalarShare>& c) {
        if (pending_) throw std::logic_error("next_triple_host after triples were read ahead into HBM: turn the prefetch off on a fabric that is driven through host-vector triples");
        prep_->next_triplet_batch(n, a, b, c);
        if (a.size() != n || b.size() != n || c.size() != n) throw std::runtime_error("preprocessing exhausted");
        next_id_ += 3 * n;
    }
    // one public Scalar as a batch of one, written by a kernel argument (no host-to-device copy, no synchronisation)
    ScalarBatch scalar_constant(const Scalar& v) {
        ScalarBatch b; b.n = 1; b.buf = DeviceBuf(eng_, ARKMPC_KIND_SCALAR, 1);
        check(ctx(), arkmpc_fill(ctx(), 1, 4, v.l, b.buf.ptr()), "fill");
        return b;
    }
    // exchange of ONE scalar both sides hold on the host (a commitment, a blinder): over the host and device links it travels as the 32
    // bytes it is; in wire mode as a ScalarBatch frame like everything else
    Scalar exchange_scalar(const Scalar& mine) {
        if (link_ == LinkMode::Wire) return exchange_values(scalar_constant(mine)).to_host()[0];
        uint64_t sid, rid;
        if (party_ == PARTY0) { sid = next_id_++; rid = next_id_++; } else { rid = next_id_++; sid = next_id_++; }
        (void)rid;
        net_->send(NetworkOutbound{sid, {mine}, {}, {}, {}});
        NetworkOutbound m = net_->receive();
        if (m.payload.size() != 1) throw std::runtime_error("MpcNetworkError: peer sent " + std::to_string(m.payload.size()) + " scalars where 1 was expected");
        return m.payload[0];
    }
    // batch_share_plaintext (fabric.rs:602-620): the sender's values become public on both sides
    ScalarBatch batch_share_plaintext(const std::vector<Scalar>& mont, size_t n, PartyId sender) {
        if (party_ == sender) { ScalarBatch b = allocate_scalars(mont); send_values(b); return b; }
        return receive_values(n);
    }
    // Point batches (NetworkPayload::PointBatch, network.rs:45-60).  Mock mode: 12 x u64 Jacobian limbs per point packed
    // into the payload vector.  Wire mode: compressed points (CurvePoint::to_bytes, curve.rs:50-55) as serde_json text; the
    // receiver decompresses with validation (CurvePoint::from_bytes, :57-63).
    template <class PB> void send_points(const PB& mine) {
        using Cv = typename PB::Curve;
        const size_t n = mine.n;
        if (link_ == LinkMode::Device) { send_device(mine.buf, Cv::PW * n); return; }
        const uint64_t id = next_id_++;
        if (!wire_) {
            std::vector<uint64_t> h = mine.to_host();
            std::vector<Scalar> pay((Cv::PW * n + 3) / 4);
            std::memcpy(pay.data(), h.data(), h.size() * 8);
            net_->send(NetworkOutbound{id, std::move(pay), {}, {}});
            return;
        }
        DeviceBuf bytes(eng_, 4 * (n ? n : 1));
        if (n) check(ctx(), Cv::to_bytes(ctx(), n, mine.buf.ptr(), reinterpret_cast<uint8_t*>(bytes.ptr())), "point to_bytes");
        size_t cap = 0, len = 0;
        arkmpc_wire_frame_bound(n, &cap);
        DeviceBuf fr(eng_, cap / 8 + 1);
        check(ctx(), arkmpc_wire_encode_bytes32(ctx(), ARKMPC_WIRE_POINT_BATCH, id, n, reinterpret_cast<const uint8_t*>(bytes.ptr()),
                                                reinterpret_cast<uint8_t*>(fr.ptr()), cap, &len), "wire_encode_bytes32");
        NetworkOutbound m{id, {}, std::vector<uint8_t>(len), {}};
        fr.download(m.frame.data(), len);
        frames_sent()++;
        net_->send(std::move(m));
    }
    // n = the point count the protocol step requires; a payload of any other size is a network error (never read past)
    template <class PB> PB receive_points(size_t n, uint64_t id = kNextId) {
        using Cv = typename PB::Curve;
        if (id == kNextId) id = next_id_++;
        NetworkOutbound m = net_->receive();
        if (link_ == LinkMode::Device) {
            if (!m.dev || m.dev->words() != Cv::PW * n) throw std::runtime_error("MpcNetworkError: unexpected point payload size");
            await_device(m);
            PB r; r.n = n; r.buf = std::move(*m.dev); r.buf.rebind(eng_); return r;
        }
        PB r; r.n = n; r.buf = DeviceBuf(eng_, ARKMPC_KIND_POINT, n);
        if (!wire_) {
            if (m.payload.size() * 4 < Cv::PW * n || m.payload.size() * 4 >= Cv::PW * n + 4) throw std::runtime_error("MpcNetworkError: unexpected point payload size");
            r.buf.upload(m.payload.data(), n * Cv::PW * 8);
            return r;
        }
        DeviceBuf fr(eng_, m.frame.size() / 8 + 2);
        fr.upload(m.frame.data(), m.frame.size());
        DeviceBuf bytes(eng_, 4 * (n ? n : 1));
        size_t cnt = 0; uint64_t rid = 0; int kind = -1;
        check(ctx(), arkmpc_wire_decode_bytes32(ctx(), reinterpret_cast<const uint8_t*>(fr.ptr()), m.frame.size(), n ? n : 1,
                                                reinterpret_cast<uint8_t*>(bytes.ptr()), &cnt, &rid, &kind), "wire_decode_bytes32");
        if (rid != id || cnt != n || kind != ARKMPC_WIRE_POINT_BATCH) throw std::runtime_error("MpcNetworkError: unexpected point frame");
        DeviceBuf okd(eng_, (n + 7) / 8 + 1);
        if (n) check(ctx(), Cv::from_bytes(ctx(), n, reinterpret_cast<const uint8_t*>(bytes.ptr()), r.buf.ptr(), reinterpret_cast<uint8_t*>(okd.ptr())),
                     "point from_bytes");
        std::vector<uint8_t> ok(n);
        okd.download(ok.data(), n);
        for (auto b : ok) if (!b) throw std::runtime_error("MpcNetworkError::SerializationError: invalid point encoding");
        return r;
    }
    template <class PB> PB exchange_points(const PB& mine) {
        if (party_ == PARTY0) { send_points(mine); return receive_points<PB>(mine.n); }
        const uint64_t rid = next_id_++;              // as exchange_values: id order of the reference, message out before blocking
        send_points(mine);
        return receive_points<PB>(mine.n, rid);
    }
    // fabric.rs:894-915.  broadcast_ok: the caller consumes the triples through column views only (the scalar Beaver kernels), so a source
    // whose batch is n copies of one value may hand out ONE record read with element stride 0 instead of n materialised copies
    void next_triple_batch(size_t n, AuthenticatedScalarBatch& a, AuthenticatedScalarBatch& b, AuthenticatedScalarBatch& c, bool broadcast_ok = false);
    // Triples of a real source are host vectors, 192 B per party-gate: over a 56 GB/s link that -- not the kernels -- bounds a circuit whose
    // operands are resident.  They therefore go up asynchronously (arkmpc_batch_from_host_async: in place over the link for split columns), and
    // with prefetch on the triples of the NEXT gate are requested from the source and started on their way as soon as this gate's K1 is enqueued,
    // i.e. under this gate's network round and K2+K3.  The source is a FIFO of triples, so reading ahead changes nothing a party can observe,
    // as long as both parties do it (they run the same program); a request of another size is served from what was read ahead, in order.
    // ARKMPC_TRIPLE_PREFETCH=0 turns it off (execute_mock_mpc).  prefetch_triples is called by batch_mul.  What is read ahead is CONSUMED from
    // the source, so (round-5 advisor finding): a source that says it cannot serve the request (triples_remaining) is not asked ahead; only the
    // source's own PreprocessingExhausted is tolerated, and asked again when the triples are really requested; triples that were consumed stay
    // with the fabric whatever happens to their upload (a failed asynchronous import is retried as a blocking one when the triples are needed,
    // and reported there).  A caller that knows its LAST multiplication calls set_triple_prefetch(false) before it: nothing is then read ahead
    // for a gate that never comes; triples already read ahead stay queued and are served first.
    void set_triple_prefetch(bool on) { prefetch_ = on; }
    bool triple_prefetch() const { return prefetch_; }
    void prefetch_triples(size_t n);
    AuthenticatedScalarBatch random_shared_scalars(size_t n);                                    // fabric.rs:917-928
    AuthenticatedScalarBatch random_shared_bits(size_t n);                                       // fabric.rs:961-984
    void random_inverse_pairs(size_t n, AuthenticatedScalarBatch& l, AuthenticatedScalarBatch& r);  // fabric.rs:942-958
    // fabric.rs:578-600
    AuthenticatedScalarBatch batch_share_scalar(const std::vector<Scalar>& vals_mont, size_t n, PartyId sender);
    AuthenticatedScalarBatch allocate_scalar_shares(const std::vector<ScalarShare>& s);
    AuthenticatedScalarBatch fill_scalar_shares(const ScalarShare& s, size_t n);     // vec![s; n] on the GPU
    AuthenticatedScalarBatch zeros_authenticated(size_t n);                          // fabric.rs:513-515: n references to the shared zero wire (0, 0)
    AuthenticatedScalarBatch ones_authenticated(size_t n);                           // fabric.rs:531-534: (party_id, mac_key), fabric.rs:234-235
    // fabric.rs:622-649: share public-format points (12 x u64 Jacobian each) held by `sender`
    template <class APB> APB batch_share_point(const std::vector<uint64_t>& points, size_t n, PartyId sender);

  private:
    PartyId party_;
    std::shared_ptr<Engine> eng_;
    std::unique_ptr<MpcNetwork> net_;
    std::unique_ptr<PreprocessingPhase> prep_;
    Scalar mac_key_;
    bool wire_ = false;
    std::vector<AuthenticatedScalarBatch> const_triple_;     // the constant source's (a, b, c) records, shared by every broadcast triple batch
    struct TripleFetch;                                       // n triples on their way to the GPU (or there already)
    std::shared_ptr<TripleFetch> fetch_triples(size_t n);
    void land(TripleFetch& t);
    std::shared_ptr<TripleFetch> pending_;                    // read ahead of need by prefetch_triples
    bool prefetch_ = true;
    int share_layout_ = ARKMPC_LAYOUT_SPLIT;
    LinkMode link_ = LinkMode::Host;
    uint64_t next_id_ = 6;   // N_CONSTANT_RESULTS (fabric.rs:55-70)
};

// Vec<AuthenticatedScalarResult<C>>: n ScalarShares resident on the GPU, in the fabric's share layout (split columns by default,
// arkworks AoS records on request).  Every op goes through the share / MAC column view of the batch handle (pointer + stride), so
// the same code drives both layouts.
class AuthenticatedScalarBatch {
  public:
    size_t n = 0;
    DeviceBuf buf;
    std::shared_ptr<MpcFabric> fabric;

    static AuthenticatedScalarBatch alloc(const std::shared_ptr<MpcFabric>& f, size_t n) {
        AuthenticatedScalarBatch r; r.n = n; r.fabric = f; r.buf = DeviceBuf(f->engine(), ARKMPC_KIND_SCALAR_SHARE, n, f->share_layout()); return r;
    }
    bool bcast = false;                                  // n logical elements, ONE stored record read with stride 0 (constant preprocessing batches)
    uint64_t* s() const { return buf.ptr(); }            // share of element i at s() + st() * i
    uint64_t* m() const { return buf.mac_ptr(); }        // MAC   of element i at m() + st() * i
    size_t st() const { return bcast ? 0 : buf.stride(); }
    bool split() const { return buf.layout() == ARKMPC_LAYOUT_SPLIT; }
    // always arkworks records, whatever the device layout (arkmpc_batch_to_host)
    std::vector<ScalarShare> to_host() const {
        std::vector<ScalarShare> v(n);
        check(ctx(*this), arkmpc_batch_to_host(ctx(*this), buf.handle(), v.data()), "batch_to_host");
        return v;
    }
    // n arkworks ScalarShare records on the device, for the entry points that take records (the point side of the ABI): the batch
    // itself if it is AoS, else a joined temporary
    struct Records {
        DeviceBuf tmp; const uint64_t* p = nullptr;
        const uint64_t* ptr() const { return p; }
    };
    Records records() const {
        Records r;
        if (!split()) { r.p = s(); return r; }
        r.tmp = DeviceBuf(fabric->engine(), ARKMPC_KIND_SCALAR_SHARE, n, ARKMPC_LAYOUT_AOS);
        if (n) check(ctx(*this), arkmpc_share_join(ctx(*this), n, s(), m(), r.tmp.ptr()), "share_join");
        r.p = r.tmp.ptr();
        return r;
    }

    // ---- linear ops (authenticated_scalar.rs:457-765) -------------------------------------------------------
    // component-wise on (share, mac): both operands are whole batches of one layout, i.e. 2n consecutive field elements each
    static AuthenticatedScalarBatch batch_add(const AuthenticatedScalarBatch& a, const AuthenticatedScalarBatch& b) {       // :457-489
        same(a, b); auto r = alloc(a.fabric, a.n); check(ctx(a), arkmpc_share_add(ctx(a), a.n, a.s(), b.s(), r.s()), "share_add"); return r;
    }
    static AuthenticatedScalarBatch batch_sub(const AuthenticatedScalarBatch& a, const AuthenticatedScalarBatch& b) {       // :662-688
        same(a, b); auto r = alloc(a.fabric, a.n); check(ctx(a), arkmpc_share_sub(ctx(a), a.n, a.s(), b.s(), r.s()), "share_sub"); return r;
    }
    static AuthenticatedScalarBatch batch_neg(const AuthenticatedScalarBatch& a) {                                          // :745-765
        auto r = alloc(a.fabric, a.n); check(ctx(a), arkmpc_share_neg(ctx(a), a.n, a.s(), r.s()), "share_neg"); return r;
    }
    static AuthenticatedScalarBatch batch_add_public(const AuthenticatedScalarBatch& a, const ScalarBatch& b) {             // :493-528
        if (a.n != b.n) throw std::invalid_argument("Cannot add batches of different sizes");
        auto r = alloc(a.fabric, a.n);
        check(ctx(a), arkmpc_share_add_public_v(ctx(a), a.n, (int)a.fabric->party_id(), a.fabric->mac_key().l, a.s(), a.m(), a.st(), b.buf.ptr(), r.s(), r.m(), r.st()), "share_add_public");
        return r;
    }
    static AuthenticatedScalarBatch batch_sub_public(const AuthenticatedScalarBatch& a, const ScalarBatch& b) {             // :691-733
        if (a.n != b.n) throw std::invalid_argument("Cannot add batches of different sizes");
        auto r = alloc(a.fabric, a.n);
        check(ctx(a), arkmpc_share_sub_public_v(ctx(a), a.n, (int)a.fabric->party_id(), a.fabric->mac_key().l, a.s(), a.m(), a.st(), b.buf.ptr(), r.s(), r.m(), r.st()), "share_sub_public");
        return r;
    }
    static AuthenticatedScalarBatch batch_mul_public(const AuthenticatedScalarBatch& a, const ScalarBatch& b) {             // :883-916
        if (a.n != b.n) throw std::invalid_argument("Cannot multiply batches of different sizes");
        auto r = alloc(a.fabric, a.n);
        check(ctx(a), arkmpc_share_mul_public_v(ctx(a), a.n, a.s(), a.m(), a.st(), b.buf.ptr(), r.s(), r.m(), r.st()), "share_mul_public");
        return r;
    }
    // batch_mul_constant (:919-949), batch_add_constant (:531-560): constants are plain Scalars the caller holds -- the public-operand
    // kernels after an upload (ScalarShare::add_public / Mul<Scalar>, share.rs:74-77, :125-131)
    static AuthenticatedScalarBatch batch_mul_constant(const AuthenticatedScalarBatch& a, const std::vector<Scalar>& consts) {
        if (a.n != consts.size()) throw std::invalid_argument("Cannot multiply batches of different sizes");
        ScalarBatch c = a.fabric->allocate_scalars(consts);
        return batch_mul_public(a, c);
    }
    static AuthenticatedScalarBatch batch_add_constant(const AuthenticatedScalarBatch& a, const std::vector<Scalar>& consts) {
        if (a.n != consts.size()) throw std::invalid_argument("Cannot add batches of different sizes");
        ScalarBatch c = a.fabric->allocate_scalars(consts);
        return batch_add_public(a, c);
    }
    // Sum for AuthenticatedScalarResult (:563-575) -> ScalarShare::sum (share.rs:103-111): one gate, a batch of ONE element
    static AuthenticatedScalarBatch sum(const AuthenticatedScalarBatch& a) {
        if (a.n == 0) throw std::invalid_argument("sum of an empty iterator");           // `values[0]` panics in the reference
        ScalarBatch t; t.n = 2; t.buf = DeviceBuf(a.fabric->engine(), ARKMPC_KIND_SCALAR, 2);
        if (a.split()) {
            check(ctx(a), arkmpc_scalar_sum(ctx(a), a.n, a.s(), t.buf.ptr()), "scalar_sum(shares)");
            check(ctx(a), arkmpc_scalar_sum(ctx(a), a.n, a.m(), t.buf.ptr() + 4), "scalar_sum(macs)");
        } else {
            check(ctx(a), arkmpc_share_sum(ctx(a), a.n, a.s(), t.buf.ptr()), "share_sum");
        }
        auto r = alloc(a.fabric, 1);                                                        // one element: (share, mac) in either layout
        check(ctx(a), arkmpc_memcpy_d2d(ctx(a), r.s(), t.buf.ptr(), 32), "d2d");
        check(ctx(a), arkmpc_memcpy_d2d(ctx(a), r.m(), t.buf.ptr() + 4, 32), "d2d");
        return r;
    }
    // ---- Beaver multiplication (:848-879) -------------------------------------------------------------------
    static AuthenticatedScalarBatch batch_mul(const AuthenticatedScalarBatch& a, const AuthenticatedScalarBatch& b) {
        same(a, b);
        const size_t n = a.n;
        auto f = a.fabric;
        if (n == 0) return alloc(f, 0);                                                       // :853-855
        AuthenticatedScalarBatch ta, tb, tc;
        f->next_triple_batch(n, ta, tb, tc, true);                                            // :859 (constant sources: one record, stride 0)
        // masked_lhs = a - beaver_a, masked_rhs = b - beaver_b, all_masks = lhs || rhs; open_batch sends `.share()` (:863-868, :141-145)
        ScalarBatch my_de; my_de.n = 2 * n; my_de.buf = DeviceBuf(f->engine(), ARKMPC_KIND_SCALAR, 2 * n);
        ScalarBatch peer_de;
        if (f->link_mode() == MpcFabric::LinkMode::Device) {
            // device link: K1 writes the payload twice -- one copy stays for this party's K2+K3, the other IS the message (no copy launch
            // in the round's dependent chain); ids and ordering as exchange_values
            auto msg = std::make_shared<DeviceBuf>(f->engine(), ARKMPC_KIND_SCALAR, 2 * n);
            check(f->ctx(), arkmpc_beaver_mask_dup(f->ctx(), n, a.s(), a.st(), b.s(), b.st(), ta.s(), ta.st(), tb.s(), tb.st(), my_de.buf.ptr(), msg->ptr()), "beaver_mask");
            f->prefetch_triples(n);                                                           // the next gate's triples start on their way under this gate's round
            peer_de = f->exchange_device_owned(std::move(msg), 8 * n, 2 * n);
        } else {
            check(f->ctx(), arkmpc_beaver_mask_v(f->ctx(), n, a.s(), a.st(), b.s(), b.st(), ta.s(), ta.st(), tb.s(), tb.st(), my_de.buf.ptr()), "beaver_mask");
            f->prefetch_triples(n);
            peer_de = f->exchange_values(my_de);                                              // the one network round (length-checked: 2n)
        }
        auto r = alloc(f, n);                                                                 // combine (:161-171) + de + d[b] + e[a] + [c] (:871-878)
        check(f->ctx(), arkmpc_beaver_finish_fused_v(f->ctx(), n, (int)f->party_id(), f->mac_key().l, my_de.buf.ptr(), peer_de.buf.ptr(),
                                                     ta.s(), ta.m(), ta.st(), tb.s(), tb.m(), tb.st(), tc.s(), tc.m(), tc.st(), r.s(), r.m(), r.st()), "beaver_finish_fused");
        return r;
    }
    // pow (:86-100): recursive squaring, element-wise over the batch; pow(0) is the fabric's SHARED ZERO wire in the reference
    // (`zero_authenticated()`, fabric.rs:507-509 -- not one), pow(1) a clone
    static AuthenticatedScalarBatch pow(const AuthenticatedScalarBatch& a, uint64_t exp) {
        if (exp == 0) return a.fabric->zeros_authenticated(a.n);
        if (exp == 1) return a.slice(0, a.n);
        AuthenticatedScalarBatch rec = pow(a, exp / 2);
        AuthenticatedScalarBatch res = batch_mul(rec, rec);
        if (exp % 2 == 1) res = batch_mul(res, a);
        return res;
    }
    // ---- inversion (:55-82): mask with a shared random r, open authenticated, invert in public, multiply back ----
    static AuthenticatedScalarBatch batch_inverse(const AuthenticatedScalarBatch& values, const Scalar& blinder, MpcError* err = nullptr) {
        if (values.n == 0) throw std::invalid_argument("cannot invert empty batch of scalars");   // assert! :60
        auto f = values.fabric;
        AuthenticatedScalarBatch r = f->random_shared_scalars(values.n);                           // step 1
        AuthenticatedScalarBatch masked = batch_mul(values, r);                                    // step 2: m_i = r_i * x_i
        AuthenticatedOpenResult opened = masked.open_authenticated_batch(blinder);
        if (err) *err = opened.err;
        ScalarBatch inv; inv.n = values.n; inv.buf = DeviceBuf(f->engine(), ARKMPC_KIND_SCALAR, values.n);         // step 3: ScalarResult::batch_inverse
        check(f->ctx(), arkmpc_scalar_batch_inverse(f->ctx(), values.n, opened.value.buf.ptr(), inv.buf.ptr()), "scalar_batch_inverse");
        return batch_mul_public(r, inv);                                                           // step 4: m_i^-1 * r_i = x_i^-1
    }
    // batch_div (:974-977): a / b = a * b^-1
    static AuthenticatedScalarBatch batch_div(const AuthenticatedScalarBatch& a, const AuthenticatedScalarBatch& b, const Scalar& blinder) {
        AuthenticatedScalarBatch b_inv = batch_inverse(b, blinder);
        return batch_mul(a, b_inv);
    }
    // ---- opening (:129-172, :278-354) -----------------------------------------------------------------------
    // the `.share()` projection a party sends (:141-145): with split columns it IS the share column -- a view that shares the storage,
    // no kernel; with AoS records an extraction pass
    ScalarBatch share_values() const {
        ScalarBatch mine; mine.n = n;
        arkmpc_ctx* c = fabric->ctx();
        if (split()) {
            arkmpc_batch* col = nullptr;
            check(c, arkmpc_batch_column(c, buf.handle(), 0, &col), "batch_column");
            mine.buf = DeviceBuf::adopt(fabric->engine(), col);
        } else {
            mine.buf = DeviceBuf(fabric->engine(), ARKMPC_KIND_SCALAR, n);
            if (n) check(c, arkmpc_share_extract(c, n, s(), mine.buf.ptr()), "share_extract");
        }
        return mine;
    }
    ScalarBatch open_batch() const {
        if (n == 0) { ScalarBatch e; e.buf = DeviceBuf(fabric->engine(), ARKMPC_KIND_SCALAR, 0); return e; }
        ScalarBatch mine = share_values();
        ScalarBatch peer = fabric->exchange_values(mine);
        ScalarBatch out; out.n = n; out.buf = DeviceBuf(fabric->engine(), ARKMPC_KIND_SCALAR, n);
        check(fabric->ctx(), arkmpc_open_combine(fabric->ctx(), n, mine.buf.ptr(), peer.buf.ptr(), out.buf.ptr()), "open_combine");
        return out;
    }
    // `blinder` is injectable for reproducible tests; the reference draws it from thread_rng (commitment.rs:67-68)
    AuthenticatedOpenResult open_authenticated_batch(const Scalar& blinder) const {
        AuthenticatedOpenResult res;
        auto f = fabric;
        if (n == 0) return res;                                                                   // :279-281
        arkmpc_ctx* c = f->ctx();
        ScalarBatch mine = share_values();
        ScalarBatch peer = f->exchange_values(mine);                                              // round 1: open_batch
        ScalarBatch opened; opened.n = n; opened.buf = DeviceBuf(f->engine(), ARKMPC_KIND_SCALAR, n);
        ScalarBatch chk; chk.n = n; chk.buf = DeviceBuf(f->engine(), ARKMPC_KIND_SCALAR, n);
        check(c, arkmpc_open_and_mac_check_v(c, n, f->mac_key().l, s(), m(), st(), peer.buf.ptr(), opened.buf.ptr(), chk.buf.ptr()), "open_and_mac_check");   // :161-171 + :299-311
        Scalar my_comm;
        check(c, arkmpc_commit_sha3(c, n, chk.buf.ptr(), blinder.l, my_comm.l), "commit_sha3");  // batch_commit (commitment.rs:63-89)
        const Scalar pc = f->exchange_scalar(my_comm);                                            // round 2: commitments
        ScalarBatch peer_chk = f->exchange_values(chk);                                           // round 3: MAC-check shares
        const Scalar pb = f->exchange_scalar(blinder);                                            // round 4: blinders
        // batch_verify_mac_check (:201-220): the peer's commitment opens correctly, and my_i + peer_i == 0 for all i
        Scalar recomputed;
        check(c, arkmpc_commit_sha3(c, n, peer_chk.buf.ptr(), pb.l, recomputed.l), "commit_sha3(verify)");
        int ok_sum = 0;
        check(c, arkmpc_mac_verify(c, n, chk.buf.ptr(), peer_chk.buf.ptr(), &ok_sum), "mac_verify");
        const bool ok = std::memcmp(recomputed.l, pc.l, 32) == 0 && ok_sum == 1;
        res.err = ok ? MpcError::None : MpcError::AuthenticationError;                            // :368-385
        res.value = std::move(opened);
        return res;
    }
    // elements [lo, lo+cnt) as a new batch (a Rust slice &v[lo..lo+cnt], materialised: whole batches stay 2n consecutive elements)
    AuthenticatedScalarBatch slice(size_t lo, size_t cnt) const {
        if (lo + cnt > n) throw std::out_of_range("slice");
        auto r = alloc(fabric, cnt);
        if (!cnt) return r;
        arkmpc_ctx* c = fabric->ctx();
        if (split()) {
            check(c, arkmpc_memcpy_d2d(c, r.s(), s() + 4 * lo, cnt * 32), "d2d");
            check(c, arkmpc_memcpy_d2d(c, r.m(), m() + 4 * lo, cnt * 32), "d2d");
        } else {
            check(c, arkmpc_memcpy_d2d(c, r.s(), s() + 8 * lo, cnt * 64), "d2d");
        }
        return r;
    }
    // iter::repeat(v[idx]).take(cnt)
    AuthenticatedScalarBatch repeat(size_t idx, size_t cnt) const {
        std::vector<ScalarShare> h = to_host();
        return fabric->fill_scalar_shares(h.at(idx), cnt);
    }
    // test helpers (authenticated_scalar.rs:1079-1111): overwrite the MAC / the share of element idx
    void modify_mac(size_t idx, const Scalar& v) { poke(idx, 1, v); }
    void modify_share(size_t idx, const Scalar& v) { poke(idx, 0, v); }

  private:
    static arkmpc_ctx* ctx(const AuthenticatedScalarBatch& a) { return a.fabric->ctx(); }
    static void same(const AuthenticatedScalarBatch& a, const AuthenticatedScalarBatch& b) {
        if (a.n != b.n) throw std::invalid_argument("Cannot operate on batches of different sizes");   // assert_eq! in the reference
        if (a.buf.layout() != b.buf.layout()) throw std::invalid_argument("batches of one fabric share one layout");
    }
    void poke(size_t idx, int half, const Scalar& v) {
        if (idx >= n) throw std::out_of_range("poke");
        check(ctx(*this), arkmpc_sync(ctx(*this)), "sync");
        check(ctx(*this), arkmpc_memcpy_h2d(ctx(*this), (half ? m() : s()) + st() * idx, v.l, 32), "h2d");
    }
};

// ---------------------------------------------------------------------------------------------------------------
// Points: CurvePointResult batches and AuthenticatedPointResult batches, generic over the curve like the reference
// (`C: CurveGroup`): PointBatchT<Bn254G1> = 12 / 24 x u64 per element, PointBatchT<Curve25519> = 16 / 32.
// ---------------------------------------------------------------------------------------------------------------
template <class Cv> struct PointBatchT {   // Vec<CurvePointResult<C>>: n public points on the GPU
    using Curve = Cv;
    size_t n = 0;
    DeviceBuf buf;
    std::vector<uint64_t> to_host() const { std::vector<uint64_t> v(Cv::PW * n); buf.download(v.data(), n * Cv::PW * 8); return v; }
};
template <class Cv> struct PointOpenResultT {   // AuthenticatedPointOpenResult (authenticated_curve.rs:286-322), one MAC-check flag per element
    std::vector<uint8_t> ok;     // 1 = commitment opened correctly and MAC shares sum to the identity
    PointBatchT<Cv> value;
    MpcError err() const { for (auto b : ok) if (!b) return MpcError::AuthenticationError; return MpcError::None; }
};

template <class Cv> class AuthenticatedPointBatchT {
  public:
    using Curve = Cv;
    using PointBatch = PointBatchT<Cv>;
    using PointOpenResult = PointOpenResultT<Cv>;
    using Self = AuthenticatedPointBatchT<Cv>;
    size_t n = 0;
    DeviceBuf buf;
    std::shared_ptr<MpcFabric> fabric;
    static Self alloc(const std::shared_ptr<MpcFabric>& f, size_t n) {
        Self r; r.n = n; r.fabric = f; r.buf = DeviceBuf(f->engine(), ARKMPC_KIND_POINT_SHARE, n); return r;
    }
    static PointBatch alloc_points(const std::shared_ptr<MpcFabric>& f, size_t n) {
        PointBatch r; r.n = n; r.buf = DeviceBuf(f->engine(), ARKMPC_KIND_POINT, n); return r;
    }
    // ---- linear ops (authenticated_curve.rs:396-621) ----
    static Self batch_add(const Self& a, const Self& b) {                                                            // :396-426
        same(a.n, b.n); auto r = alloc(a.fabric, a.n); check(c(a), Cv::share_add(c(a), a.n, a.buf.ptr(), b.buf.ptr(), r.buf.ptr()), "pointshare_add"); return r;
    }
    static Self batch_sub(const Self& a, const Self& b) {                                                            // :520-550
        same(a.n, b.n); auto r = alloc(a.fabric, a.n); check(c(a), Cv::share_sub(c(a), a.n, a.buf.ptr(), b.buf.ptr(), r.buf.ptr()), "pointshare_sub"); return r;
    }
    static Self batch_neg(const Self& a) {                                                                           // :604-621
        auto r = alloc(a.fabric, a.n); check(c(a), Cv::share_neg(c(a), a.n, a.buf.ptr(), r.buf.ptr()), "pointshare_neg"); return r;
    }
    static Self batch_add_public(const Self& a, const PointBatch& b) {                                               // :429-463
        same(a.n, b.n); auto r = alloc(a.fabric, a.n);
        check(c(a), Cv::share_add_public(c(a), a.n, (int)a.fabric->party_id(), a.fabric->mac_key().l, a.buf.ptr(), b.buf.ptr(), r.buf.ptr()), "pointshare_add_public");
        return r;
    }
    static Self batch_sub_public(const Self& a, const PointBatch& b) {                                               // :553-575 (curve/share.rs:63-65)
        same(a.n, b.n); auto r = alloc(a.fabric, a.n);
        check(c(a), Cv::share_sub_public(c(a), a.n, (int)a.fabric->party_id(), a.fabric->mac_key().l, a.buf.ptr(), b.buf.ptr(), r.buf.ptr()), "pointshare_sub_public");
        return r;
    }
    static Self batch_mul_public(const ScalarBatch& s, const Self& b) {                                              // :718-751
        same(s.n, b.n); auto r = alloc(b.fabric, b.n); check(c(b), Cv::share_mul_public(c(b), b.n, b.buf.ptr(), s.buf.ptr(), r.buf.ptr()), "pointshare_mul_public"); return r;
    }
    static Self batch_mul_generator(const AuthenticatedScalarBatch& a) {                                              // :754-780
        auto r = alloc(a.fabric, a.n); auto ar = a.records();
        check(a.fabric->ctx(), Cv::scalarshare_mul_generator(a.fabric->ctx(), a.n, ar.ptr(), r.buf.ptr()), "mul_generator"); return r;
    }
    // CurvePointResult::batch_mul (curve.rs:459-479) and batch_mul_authenticated (curve.rs:483-517)
    static PointBatch point_batch_mul(const std::shared_ptr<MpcFabric>& f, const ScalarBatch& s, const PointBatch& p) {
        same(s.n, p.n); auto r = alloc_points(f, p.n); check(f->ctx(), Cv::scalar_mul(f->ctx(), p.n, p.buf.ptr(), s.buf.ptr(), r.buf.ptr()), "scalar_mul"); return r;
    }
    static Self batch_mul_authenticated(const AuthenticatedScalarBatch& a, const PointBatch& p) {
        same(a.n, p.n); auto r = alloc(a.fabric, a.n); auto ar = a.records();
        check(a.fabric->ctx(), Cv::scalarshare_mul_point(a.fabric->ctx(), a.n, ar.ptr(), p.buf.ptr(), r.buf.ptr()), "scalarshare_mul_point"); return r;
    }
    // ---- opening (:66-109) ----
    PointBatch open_batch() const {
        auto f = fabric;
        PointBatch mine = alloc_points(f, n);
        if (n == 0) return mine;
        check(f->ctx(), Cv::share_extract(f->ctx(), n, buf.ptr(), mine.buf.ptr()), "pointshare_extract");
        PointBatch peer = f->exchange_points(mine);
        PointBatch out = alloc_points(f, n);
        check(f->ctx(), Cv::add(f->ctx(), n, mine.buf.ptr(), peer.buf.ptr(), out.buf.ptr()), "point add");
        return out;
    }
    // :190-283 -- per-element commitments (n separate SHA3 commits, :227), three exchanges, per-element verification
    PointOpenResult open_authenticated_batch(const std::vector<Scalar>& blinders_mont) const {
        PointOpenResult res;
        auto f = fabric;
        if (n == 0) return res;
        if (blinders_mont.size() != n) throw std::invalid_argument("one blinder per element");
        arkmpc_ctx* cx = f->ctx();
        PointBatch opened = open_batch();
        PointBatch chk = alloc_points(f, n);                                                       // value*mac_key - mac (:215-220)
        check(cx, Cv::mac_check_shares(cx, n, f->mac_key().l, opened.buf.ptr(), buf.ptr(), chk.buf.ptr()), "point_mac_check_shares");
        ScalarBatch bl = f->allocate_scalars(blinders_mont);
        ScalarBatch comm; comm.n = n; comm.buf = DeviceBuf(f->engine(), ARKMPC_KIND_SCALAR, n);
        check(cx, Cv::commit_points(cx, n, chk.buf.ptr(), bl.buf.ptr(), comm.buf.ptr()), "commit_points_sha3");
        ScalarBatch peer_comm = f->exchange_values(comm);                                          // commitments   (length-checked: n)
        PointBatch peer_chk = f->exchange_points(chk);                                             // MAC-check points
        ScalarBatch peer_bl = f->exchange_values(bl);                                              // blinders
        ScalarBatch recomputed; recomputed.n = n; recomputed.buf = DeviceBuf(f->engine(), ARKMPC_KIND_SCALAR, n);
        check(cx, Cv::commit_points(cx, n, peer_chk.buf.ptr(), peer_bl.buf.ptr(), recomputed.buf.ptr()), "commit_points_sha3(verify)");
        DeviceBuf okd(f->engine(), (n + 7) / 8 + 1);
        check(cx, Cv::mac_verify(cx, n, chk.buf.ptr(), peer_chk.buf.ptr(), reinterpret_cast<uint8_t*>(okd.ptr())), "point_mac_verify");
        res.ok.resize(n);
        okd.download(res.ok.data(), n);
        std::vector<Scalar> rc = recomputed.to_host(), pc = peer_comm.to_host();
        for (size_t i = 0; i < n; ++i) if (std::memcmp(rc[i].l, pc[i].l, 32) != 0) res.ok[i] = 0;  // verify_mac_check (:112-138)
        res.value = std::move(opened);
        return res;
    }
    // ---- Beaver point x shared-scalar multiplication (:682-714): [x * yG] = deG + d[bG] + [a]eG + [c]G ----
    // The reference evaluates the four terms as written: 6 variable-base and 4 generator scalar-muls per element and party
    // (deG: 1, d[bG]: 2, [a]eG: 2, mac_key * deG inside add_public: 1; [b]G and [c]G: 2 + 2).  Every term is a multiple of either the
    // opened point eG or of G, and [bG] = [b]G, so by distributivity -- exact in the group, share by share and MAC by MAC --
    //     deG + d[bG] + [a]eG + [c]G  =  ([a] + d) * eG  +  ([c] + d[b]) * G
    // where "[a] + d" is ScalarShare::add_public (share += d iff PARTY0, mac += mac_key * d: what add_public(dbG, deG) does to the points,
    // share.rs:74-77 / curve/share.rs:57-60).  That is 2 variable-base + 4 generator scalar-muls (the generator ones on the fixed-base
    // table): the form this engine runs.  `literal` = the reference's op sequence, kept for the parity test (both must open to the same
    // points and pass the same MAC check; mock_mpc scenario point_mul_forms).
    static Self batch_mul(const AuthenticatedScalarBatch& a, const Self& b, bool literal = point_mul_literal()) {
        same(a.n, b.n);
        if (a.n == 0) return alloc(a.fabric, 0);
        AuthenticatedScalarBatch ta, tb, tc;
        a.fabric->next_triple_batch(a.n, ta, tb, tc);
        return batch_mul_with_triple(a, b, ta, tb, tc, literal);
    }
    // the gate itself, on a triple the caller has drawn (the parity test evaluates both forms on ONE triple)
    static Self batch_mul_with_triple(const AuthenticatedScalarBatch& a, const Self& b, const AuthenticatedScalarBatch& ta,
                                      const AuthenticatedScalarBatch& tb, const AuthenticatedScalarBatch& tc, bool literal) {
        same(a.n, b.n);
        const size_t n = a.n;
        auto f = a.fabric;
        Self beaver_b_gen = batch_mul_generator(tb);                                               // :696
        AuthenticatedScalarBatch masked_rhs = AuthenticatedScalarBatch::batch_sub(a, ta);          // :698
        Self masked_lhs = batch_sub(b, beaver_b_gen);                                              // :699
        PointBatch eG_open = masked_lhs.open_batch();                                              // :701
        ScalarBatch d_open = masked_rhs.open_batch();                                              // :702
        if (!literal) {                                                                            // one gate of the C ABI: the point-side K3
            Self r = alloc(f, n);
            auto ra = ta.records(), rb = tb.records(), rc = tc.records();            // the point side of the ABI takes arkworks ScalarShare records
            check(f->ctx(), Cv::beaver_finish(f->ctx(), n, (int)f->party_id(), f->mac_key().l, d_open.buf.ptr(), eG_open.buf.ptr(), ra.ptr(),
                                              rb.ptr(), rc.ptr(), r.buf.ptr()), "point_beaver_finish");
            return r;
        }
        PointBatch deG = point_batch_mul(f, d_open, eG_open);                                      // :705
        Self dbG = batch_mul_public(d_open, beaver_b_gen);                                         // :706
        Self aeG = batch_mul_authenticated(ta, eG_open);                                           // :707
        Self cG = batch_mul_generator(tc);                                                         // :708
        Self de_db_G = batch_add_public(dbG, deG);                                                 // :710
        Self ae_c_G = batch_add(aeG, cG);                                                          // :711
        return batch_add(de_db_G, ae_c_G);                                                         // :713
    }
    static bool point_mul_literal() {
        static const bool v = std::getenv("ARKMPC_POINT_MUL_LITERAL") && std::getenv("ARKMPC_POINT_MUL_LITERAL")[0] == '1';
        return v;
    }

    // ---- multiscalar multiplication (:787-806): batch_mul, then one gate summing the PointShares ----
    // With the regrouped batch_mul above, sum_i [x_i * Y_i] = sum_i ([a_i] + d_i) * eG_i + sum_i ([c_i] + d_i [b_i]) * G: the first sum is an
    // authenticated MSM over the opened points (the bucket method on BN254, arkmpc_g1_msm_authenticated), the second a sum of generator
    // multiples.  Same group elements, share by share; `literal` = batch_mul + sum as the reference writes it.
    static Self msm(const AuthenticatedScalarBatch& scalars, const Self& points, bool literal = point_mul_literal()) {
        if (scalars.n != points.n) throw std::invalid_argument("multiscalar_mul requires equal length vectors");
        if (scalars.n == 0) throw std::invalid_argument("multiscalar_mul requires non-empty vectors");
        const size_t n = scalars.n;
        auto f = scalars.fabric;
        if (literal) {
            Self prod = batch_mul(scalars, points, true);
            Self r = alloc(f, 1);
            check(c(points), Cv::share_sum(c(points), prod.n, prod.buf.ptr(), r.buf.ptr()), "pointshare_sum");
            return r;
        }
        AuthenticatedScalarBatch ta, tb, tc;
        f->next_triple_batch(n, ta, tb, tc);
        Self beaver_b_gen = batch_mul_generator(tb);                                               // :696
        AuthenticatedScalarBatch masked_rhs = AuthenticatedScalarBatch::batch_sub(scalars, ta);    // :698
        Self masked_lhs = batch_sub(points, beaver_b_gen);                                         // :699
        PointBatch eG_open = masked_lhs.open_batch();                                              // :701
        ScalarBatch d_open = masked_rhs.open_batch();                                              // :702
        AuthenticatedScalarBatch on_eG = AuthenticatedScalarBatch::batch_add_public(ta, d_open);
        AuthenticatedScalarBatch on_G = AuthenticatedScalarBatch::batch_add(tc, AuthenticatedScalarBatch::batch_mul_public(tb, d_open));
        Self over_eG = msm_authenticated(on_eG, eG_open);
        Self gen = batch_mul_generator(on_G);
        Self over_G = alloc(f, 1);
        check(f->ctx(), Cv::share_sum(f->ctx(), n, gen.buf.ptr(), over_G.buf.ptr()), "pointshare_sum");
        return batch_add(over_eG, over_G);
    }
    // CurvePoint::msm / CurvePointResult::msm_results (curve.rs:549-560, :588-603): public scalars x public points -> one point
    static PointBatch point_msm(const std::shared_ptr<MpcFabric>& f, const ScalarBatch& scalars, const PointBatch& points) {
        if (scalars.n != points.n) throw std::invalid_argument("msm cannot compute on vectors of unequal length");
        auto r = alloc_points(f, 1);
        check(f->ctx(), Cv::msm(f->ctx(), points.n, points.buf.ptr(), scalars.buf.ptr(), r.buf.ptr()), "msm");
        return r;
    }
    // CurvePointResult::msm_authenticated (curve.rs:618-642 / :701-731): authenticated scalars x public points; a local
    // gate -- PointShare(msm(shares, P), msm(macs, P))
    static Self msm_authenticated(const AuthenticatedScalarBatch& scalars, const PointBatch& points) {
        if (scalars.n != points.n) throw std::invalid_argument("msm cannot compute on vectors of unequal length");
        auto r = alloc(scalars.fabric, 1);
        auto sr = scalars.records();
        check(scalars.fabric->ctx(), Cv::msm_authenticated(scalars.fabric->ctx(), points.n, points.buf.ptr(), sr.ptr(), r.buf.ptr()), "msm_authenticated");
        return r;
    }

  private:
    static arkmpc_ctx* c(const Self& a) { return a.fabric->ctx(); }
    static void same(size_t a, size_t b) { if (a != b) throw std::invalid_argument("Cannot operate on batches of different sizes"); }
};
using PointBatch = PointBatchT<Bn254G1>;
using PointOpenResult = PointOpenResultT<Bn254G1>;
using AuthenticatedPointBatch = AuthenticatedPointBatchT<Bn254G1>;
using EdPointBatch = PointBatchT<Curve25519>;
using EdPointOpenResult = PointOpenResultT<Curve25519>;
using AuthenticatedEdPointBatch = AuthenticatedPointBatchT<Curve25519>;

// fabric.rs:622-649: the sender broadcasts val - mask*G; both sides compute mask_share*G + masked (batch_mul_generator, batch_add_public)
template <class APB>
inline APB MpcFabric::batch_share_point(const std::vector<uint64_t>& points, size_t n, PartyId sender) {
    using Cv = typename APB::Curve;
    using PB = typename APB::PointBatch;
    auto self = shared_from_this();
    PB masked;
    std::vector<ScalarShare> mask_shares;
    if (party_ == sender) {
        if (points.size() != Cv::PW * n) throw std::invalid_argument("batch_share_point: n points expected");
        auto lm = prep_->next_local_input_mask_batch(n);
        ScalarBatch masks = allocate_scalars(lm.first);
        PB mg = APB::alloc_points(self, n), vals = APB::alloc_points(self, n);
        vals.buf.upload(points.data(), n * Cv::PW * 8);
        if (n) check(ctx(), Cv::generator_mul(ctx(), n, masks.buf.ptr(), mg.buf.ptr()), "generator_mul");
        masked = APB::alloc_points(self, n);
        if (n) check(ctx(), Cv::sub(ctx(), n, vals.buf.ptr(), mg.buf.ptr(), masked.buf.ptr()), "point sub");
        send_points(masked);                                // plaintext broadcast of the masked points (batch_share_plaintext)
        mask_shares = std::move(lm.second);
    } else {
        mask_shares = prep_->next_counterparty_input_mask_batch(n);
        masked = receive_points<PB>(n);
    }
    AuthenticatedScalarBatch shares = allocate_scalar_shares(mask_shares);
    APB masks_g = APB::batch_mul_generator(shares);
    return APB::batch_add_public(masks_g, masked);
}

inline AuthenticatedScalarBatch MpcFabric::allocate_scalar_shares(const std::vector<ScalarShare>& s) {
    AuthenticatedScalarBatch r; r.n = s.size(); r.fabric = shared_from_this();
    arkmpc_batch* b = nullptr;
    check(ctx(), arkmpc_batch_from_host(ctx(), ARKMPC_KIND_SCALAR_SHARE, share_layout_, s.size(), s.data(), &b), "batch_from_host");
    r.buf = DeviceBuf::adopt(eng_, b);
    return r;
}
inline AuthenticatedScalarBatch MpcFabric::fill_scalar_shares(const ScalarShare& s, size_t n) {
    auto r = AuthenticatedScalarBatch::alloc(shared_from_this(), n);
    if (r.split()) {
        check(ctx(), arkmpc_fill(ctx(), n, 4, s.share.l, r.s()), "fill");
        check(ctx(), arkmpc_fill(ctx(), n, 4, s.mac.l, r.m()), "fill");
    } else {
        check(ctx(), arkmpc_fill(ctx(), n, 8, s.share.l, r.s()), "fill");
    }
    return r;
}
inline AuthenticatedScalarBatch MpcFabric::zeros_authenticated(size_t n) { return fill_scalar_shares(ScalarShare{Scalar{{0, 0, 0, 0}}, Scalar{{0, 0, 0, 0}}}, n); }
inline AuthenticatedScalarBatch MpcFabric::ones_authenticated(size_t n) { return fill_scalar_shares(ScalarShare{eng_->from_u64(party_), mac_key_}, n); }
struct MpcFabric::TripleFetch {
    size_t n = 0;
    std::vector<ScalarShare> h[3];                           // the source's Vecs, alive until the import has read them (empty: lent in place)
    AuthenticatedScalarBatch t[3];
    bool up[3] = {false, false, false};                      // t[k]'s import was started; false = the records wait in h[k] for land() to import them
    bool landed = false;
};
inline void MpcFabric::next_triple_batch(size_t n, AuthenticatedScalarBatch& a, AuthenticatedScalarBatch& b, AuthenticatedScalarBatch& c, bool broadcast_ok) {
    ScalarShare ca, cb, cc;
    if (prep_->constant_triplet(ca, cb, cc)) {
        next_id_ += 3 * n;
        if (broadcast_ok) {                       // `vec![share; n]` (offline_prep.rs:137-158) as one record + stride 0, written once per fabric
            if (const_triple_.empty() || const_triple_[0].buf.layout() != share_layout_) {
                const_triple_.clear();
                const_triple_.push_back(fill_scalar_shares(ca, 1)); const_triple_.push_back(fill_scalar_shares(cb, 1)); const_triple_.push_back(fill_scalar_shares(cc, 1));
                for (auto& t : const_triple_) t.fabric.reset();           // no cycle fabric -> batch -> fabric
            }
            AuthenticatedScalarBatch* out[3] = {&a, &b, &c};
            for (int k = 0; k < 3; ++k) {
                AuthenticatedScalarBatch r; r.n = n; r.bcast = true; r.fabric = shared_from_this(); r.buf = const_triple_[k].buf.share();
                *out[k] = std::move(r);
            }
            return;
        }
        a = fill_scalar_shares(ca, n); b = fill_scalar_shares(cb, n); c = fill_scalar_shares(cc, n);
        return;
    }
    AuthenticatedScalarBatch* out[3] = {&a, &b, &c};
    if (pending_ && pending_->n != n) {
        // a request of another size than what was read ahead: the source is a FIFO, so serve from the read-ahead triples first, in order
        land(*pending_);
        std::shared_ptr<TripleFetch> p = std::move(pending_);
        pending_.reset();
        if (n < p->n) {
            auto rest = std::make_shared<TripleFetch>();
            rest->n = p->n - n;
            for (int k = 0; k < 3; ++k) { *out[k] = p->t[k].slice(0, n); rest->t[k] = p->t[k].slice(n, p->n - n); rest->t[k].fabric.reset(); rest->up[k] = true; }
            pending_ = std::move(rest);
        } else {
            std::shared_ptr<TripleFetch> more = fetch_triples(n - p->n);
            land(*more);
            for (int k = 0; k < 3; ++k) {
                AuthenticatedScalarBatch r = AuthenticatedScalarBatch::alloc(shared_from_this(), n);
                const AuthenticatedScalarBatch* part[2] = {&p->t[k], &more->t[k]};
                size_t lo = 0;
                for (const AuthenticatedScalarBatch* q : part) {
                    if (!q->n) continue;
                    if (r.split()) {
                        check(ctx(), arkmpc_memcpy_d2d(ctx(), r.s() + 4 * lo, q->s(), q->n * 32), "d2d");
                        check(ctx(), arkmpc_memcpy_d2d(ctx(), r.m() + 4 * lo, q->m(), q->n * 32), "d2d");
                    } else check(ctx(), arkmpc_memcpy_d2d(ctx(), r.s() + 8 * lo, q->s(), q->n * 64), "d2d");
                    lo += q->n;
                }
                *out[k] = std::move(r);
            }
        }
        next_id_ += 3 * n;
        return;
    }
    if (pending_) land(*pending_);                           // (in place: if this throws, the consumed triples are still the fabric's)
    std::shared_ptr<TripleFetch> t = pending_ ? std::move(pending_) : fetch_triples(n);
    pending_.reset();
    land(*t);
    next_id_ += 3 * n;
    for (int k = 0; k < 3; ++k) *out[k] = std::move(t->t[k]);
}
// n triples from the source, started on their way to the GPU; returns at once for vectors that can go up asynchronously.  Throws only what the
// SOURCE throws (nothing consumed).  Once the source has handed the triples over they are consumed -- this party's FIFO has advanced, as its
// peer's will -- so from then on nothing is dropped: an import that could not be started leaves its records in the fetch (a lent range is copied
// out: it is only good until the next call on the source) and land() imports them, or reports why it cannot, when the triples are needed.
inline std::shared_ptr<MpcFabric::TripleFetch> MpcFabric::fetch_triples(size_t n) {
    auto f = std::make_shared<TripleFetch>();
    f->n = n;
    const ScalarShare* p[3] = {nullptr, nullptr, nullptr};
    if (!prep_->borrow_triplet_batch(n, &p[0], &p[1], &p[2])) {
        prep_->next_triplet_batch(n, f->h[0], f->h[1], f->h[2]);
        for (int k = 0; k < 3; ++k) {
            if (f->h[k].size() != n) throw PreprocessingExhausted("preprocessing exhausted");   // structs.rs:189 asserts
            p[k] = f->h[k].data();
        }
    }
    for (int k = 0; k < 3; ++k) {
        f->t[k].n = n;
        arkmpc_batch* b = nullptr;
        if (arkmpc_batch_from_host_async(ctx(), ARKMPC_KIND_SCALAR_SHARE, share_layout_, n, p[k], &b) == ARKMPC_OK) {
            f->t[k].buf = DeviceBuf::adopt(eng_, b);         // (no fabric pointer while the fetch may be parked in pending_: no cycle fabric -> batch -> fabric)
            f->up[k] = true;
        } else if (f->h[k].empty()) f->h[k].assign(p[k], p[k] + n);
    }
    return f;
}
// the compute stream waits for the imports; the source's memory is released (blocks until the uploads have read it -- long done for triples
// that were read ahead a gate ago).  Records whose asynchronous import could not be started go up now, blocking; a failure here throws with the
// records still in the fetch.
inline void MpcFabric::land(TripleFetch& t) {
    if (t.landed) return;
    for (int k = 0; k < 3; ++k) {
        if (!t.up[k]) {
            arkmpc_batch* b = nullptr;
            check(ctx(), arkmpc_batch_from_host(ctx(), ARKMPC_KIND_SCALAR_SHARE, share_layout_, t.n, t.h[k].data(), &b), "batch_from_host (triples whose asynchronous import failed)");
            t.t[k].buf = DeviceBuf::adopt(eng_, b);
            t.up[k] = true;
        }
    }
    for (int k = 0; k < 3; ++k) {
        check(ctx(), arkmpc_batch_acquire(ctx(), t.t[k].buf.handle()), "batch_acquire");
        check(ctx(), arkmpc_batch_host_release(ctx(), t.t[k].buf.handle()), "batch_host_release");
        t.t[k].fabric = shared_from_this();
        std::vector<ScalarShare>().swap(t.h[k]);
    }
    t.landed = true;
}
inline void MpcFabric::prefetch_triples(size_t n) {
    if (!prefetch_ || pending_ || !n) return;
    ScalarShare ca, cb, cc;
    if (prep_->constant_triplet(ca, cb, cc)) return;         // constant sources are one record on the GPU already
    if (prep_->triples_remaining() < n) return;              // the source cannot serve a gate of this size any more: nothing is consumed for one that may never come
    try { pending_ = fetch_triples(n); } catch (const PreprocessingExhausted&) { pending_.reset(); }      // (thrown before anything was consumed; asked again at need)
}
inline void MpcFabric::random_inverse_pairs(size_t n, AuthenticatedScalarBatch& l, AuthenticatedScalarBatch& r) {
    std::vector<ScalarShare> hl, hr;
    prep_->next_shared_inverse_pair_batch(n, hl, hr);
    next_id_ += 2 * n;
    l = allocate_scalar_shares(hl); r = allocate_scalar_shares(hr);
}
inline AuthenticatedScalarBatch MpcFabric::random_shared_bits(size_t n) {
    std::vector<ScalarShare> v = prep_->next_shared_bit_batch(n);
    if (v.size() != n) throw std::runtime_error("preprocessing exhausted");
    next_id_ += n;
    return allocate_scalar_shares(v);
}
inline AuthenticatedScalarBatch MpcFabric::random_shared_scalars(size_t n) {
    std::vector<ScalarShare> v = prep_->next_shared_value_batch(n);
    next_id_ += n;
    return allocate_scalar_shares(v);
}
// fabric.rs:578-600: the sender broadcasts val - mask; both sides do mask_share.add_public(masked)
inline AuthenticatedScalarBatch MpcFabric::batch_share_scalar(const std::vector<Scalar>& vals_mont, size_t n, PartyId sender) {
    ScalarBatch masked;
    std::vector<ScalarShare> mask_shares;
    Scalar cv; ScalarShare cl, cc;
    if (prep_->constant_input_masks(cv, cl, cc)) {            // masks described by one value: filled on the GPU
        if (party_ == sender) {
            ScalarBatch vals = allocate_scalars(vals_mont), masks;
            masks.n = n; masks.buf = DeviceBuf(eng_, ARKMPC_KIND_SCALAR, n);
            check(ctx(), arkmpc_fill(ctx(), n, 4, cv.l, masks.buf.ptr()), "fill");
            masked.n = n; masked.buf = DeviceBuf(eng_, ARKMPC_KIND_SCALAR, n);
            if (n) check(ctx(), arkmpc_scalar_sub(ctx(), n, vals.buf.ptr(), masks.buf.ptr(), masked.buf.ptr()), "scalar_sub");
            send_values(masked);
        } else {
            masked = receive_values(n);
        }
        return AuthenticatedScalarBatch::batch_add_public(fill_scalar_shares(party_ == sender ? cl : cc, n), masked);
    }
    if (party_ == sender) {
        auto lm = prep_->next_local_input_mask_batch(n);
        ScalarBatch vals = allocate_scalars(vals_mont), masks = allocate_scalars(lm.first);
        masked.n = n; masked.buf = DeviceBuf(eng_, ARKMPC_KIND_SCALAR, n);
        if (n) check(ctx(), arkmpc_scalar_sub(ctx(), n, vals.buf.ptr(), masks.buf.ptr(), masked.buf.ptr()), "scalar_sub");
        send_values(masked);
        mask_shares = std::move(lm.second);
    } else {
        mask_shares = prep_->next_counterparty_input_mask_batch(n);
        masked = receive_values(n);
    }
    AuthenticatedScalarBatch shares = allocate_scalar_shares(mask_shares);
    return AuthenticatedScalarBatch::batch_add_public(shares, masked);
}

// ---------------------------------------------------------------------------------------------------------------
// One party over SEVERAL GPUs: the multi-device group of the C ABI (arkmpc_group_*) under the same fabric.  A party of the reference
// is one process (fabric.rs:402-466), so this is the shape an 8-GPU deployment has: the fabric (party id, MAC key share, preprocessing,
// network, result ids) is the one above; batches are range-sharded over the group's members -- member g owns elements
// [g*n/G, (g+1)*n/G) -- and the scalar hot path (batch_mul, open_authenticated_batch) runs as the group entry points.  Payloads cross
// the link as host vectors: every member DMAs its range to / from host memory over its own PCIe link (arkmpc_group_gather_d2h /
// _scatter_h2d), which is where a NIC would pick them up.
// ---------------------------------------------------------------------------------------------------------------
class GroupFabric;
struct GroupOpenResult {            // AuthenticatedScalarOpenResult for a sharded batch: the opened values gathered to the host
    MpcError err = MpcError::None;
    std::vector<Scalar> value;
};
// a sharded vector: `segs` segments of `ew` words per element on every member (see include/arkmpc.h); freed with the group
class ShardedBuf {
  public:
    ShardedBuf() = default;
    ShardedBuf(arkmpc_group* g, size_t n, size_t segs, size_t ew) : g_(g), n_(n), segs_(segs), ew_(ew), p_((size_t)arkmpc_group_size(g), nullptr) {
        if (arkmpc_group_malloc(g, n, segs, ew, p_.data()) != ARKMPC_OK) throw std::runtime_error(std::string("arkmpc_group_malloc: ") + arkmpc_group_last_error(g));
    }
    ~ShardedBuf() { if (g_) arkmpc_group_free(g_, p_.data()); }
    ShardedBuf(ShardedBuf&& o) noexcept { *this = std::move(o); }
    ShardedBuf& operator=(ShardedBuf&& o) noexcept {
        if (this != &o) { if (g_) arkmpc_group_free(g_, p_.data()); g_ = o.g_; n_ = o.n_; segs_ = o.segs_; ew_ = o.ew_; p_ = std::move(o.p_); o.g_ = nullptr; }
        return *this;
    }
    ShardedBuf(const ShardedBuf&) = delete;
    uint64_t* const* ptrs() const { return p_.data(); }
    const uint64_t* const* cptrs() const { return const_cast<const uint64_t* const*>(p_.data()); }
    size_t n() const { return n_; }

  private:
    arkmpc_group* g_ = nullptr;
    size_t n_ = 0, segs_ = 0, ew_ = 0;
    std::vector<uint64_t*> p_;
};
struct ShardedShares {              // Vec<AuthenticatedScalarResult<C>> range-sharded over the group, in the fabric's share layout
    size_t n = 0;
    std::shared_ptr<GroupFabric> fabric;   // declared BEFORE buf: members die in reverse order, and buf's destructor frees through the group this keeps alive
    ShardedBuf buf;
};
class GroupFabric : public std::enable_shared_from_this<GroupFabric> {
  public:
    GroupFabric(std::shared_ptr<MpcFabric> fabric, const std::vector<int>& device_ids) : f_(std::move(fabric)) {
        f_->set_triple_prefetch(false);                      // this fabric's triples are taken as host vectors (next_triple_host): no reading ahead into ONE device's HBM
        if (arkmpc_group_create(f_->engine()->field_id(), (int)device_ids.size(), device_ids.data(), &g_) != ARKMPC_OK)
            throw std::runtime_error("arkmpc_group_create failed: the engine needs GPUs, there is no CPU fallback");
    }
    ~GroupFabric() {
        for (auto* q : const_trip_) if (q) arkmpc_host_free(q);
        if (g_) arkmpc_group_destroy(g_);
    }
    GroupFabric(const GroupFabric&) = delete;
    arkmpc_group* group() const { return g_; }
    int layout() const { return f_->share_layout(); }
    std::shared_ptr<MpcFabric> fabric() const { return f_; }
    void gcheck(int rc, const char* what) const { if (rc != ARKMPC_OK) throw std::runtime_error(std::string(what) + ": " + arkmpc_group_last_error(g_)); }
    ShardedShares alloc(size_t n) {
        ShardedShares r; r.n = n; r.fabric = shared_from_this();
        r.buf = layout() == ARKMPC_LAYOUT_SPLIT ? ShardedBuf(g_, n, 2, 4) : ShardedBuf(g_, n, 1, 8);
        return r;
    }
    ShardedShares shares_from_host(const std::vector<ScalarShare>& v) {
        ShardedShares r = alloc(v.size());
        gcheck(arkmpc_group_shares_from_host(g_, layout(), v.size(), v.empty() ? nullptr : &v[0].share.l[0], r.buf.ptrs()), "group_shares_from_host");
        return r;
    }
    std::vector<ScalarShare> to_host(const ShardedShares& a) {
        std::vector<ScalarShare> v(a.n);
        gcheck(arkmpc_group_shares_to_host(g_, layout(), a.n, a.buf.cptrs(), a.n ? &v[0].share.l[0] : nullptr), "group_shares_to_host");
        return v;
    }
    // Payload vectors are RECYCLED: a fresh 64 MiB std::vector costs 20-25 ms before the first useful byte (mmap, 16 K page faults, zero
    // fill) -- three times the gate it carries.  Every exchange gives one vector away (it moves into the message) and gets one of the same
    // size back (the peer's), so the received vector, once consumed, is the next payload of that size: after the first round of a given size
    // no exchange allocates.  (A Rust caller gets the same from a Vec pool or the Pinned allocator of INTEGRATION.md section 2a.)
    std::vector<Scalar> take_scalars(size_t count) {
        for (size_t i = 0; i < spare_.size(); ++i)
            if (spare_[i].size() == count) { std::vector<Scalar> v = std::move(spare_[i]); spare_.erase(spare_.begin() + i); return v; }
        return std::vector<Scalar>(count);
    }
    void give_back(std::vector<Scalar>&& v) {
        if (v.empty()) return;
        if (spare_.size() >= 8) spare_.erase(spare_.begin());
        spare_.push_back(std::move(v));
    }
    // the d||e / share-value / MAC-check exchanges: range DMAs to the host, the message, range DMAs back (ids as MpcFabric::exchange_values)
    ShardedBuf exchange(const ShardedBuf& mine, size_t n, size_t segs) {
        std::vector<Scalar> pay = take_scalars(n * segs);
        gcheck(arkmpc_group_gather_d2h(g_, n, segs, 4, mine.cptrs(), n ? &pay[0].l[0] : nullptr), "group_gather_d2h");
        std::vector<Scalar> peer = f_->exchange_host_values(std::move(pay));
        ShardedBuf r(g_, n, segs, 4);
        gcheck(arkmpc_group_scatter_h2d(g_, n, segs, 4, n ? &peer[0].l[0] : nullptr, r.ptrs()), "group_scatter_h2d");
        give_back(std::move(peer));
        return r;
    }
    // Beaver multiplication over the group (authenticated_scalar.rs:848-879)
    ShardedShares batch_mul(const ShardedShares& a, const ShardedShares& b) {
        if (a.n != b.n) throw std::invalid_argument("Cannot operate on batches of different sizes");
        const size_t n = a.n;
        if (n == 0) return alloc(0);
        std::vector<ScalarShare> ha, hb, hc;
        f_->next_triple_host(n, ha, hb, hc);
        ShardedShares ta = shares_from_host(ha), tb = shares_from_host(hb), tc = shares_from_host(hc);
        ShardedBuf my_de(g_, n, 2, 4);
        gcheck(arkmpc_group_beaver_mask(g_, layout(), n, a.buf.cptrs(), b.buf.cptrs(), ta.buf.cptrs(), tb.buf.cptrs(), my_de.ptrs()), "group_beaver_mask");
        ShardedBuf peer_de = exchange(my_de, n, 2);
        ShardedShares r = alloc(n);
        gcheck(arkmpc_group_beaver_finish_fused(g_, layout(), n, (int)f_->party_id(), f_->mac_key().l, my_de.cptrs(), peer_de.cptrs(), ta.buf.cptrs(),
                                                tb.buf.cptrs(), tc.buf.cptrs(), r.buf.ptrs()), "group_beaver_finish_fused");
        return r;                                             // the temporaries' frees are stream-ordered per member (arkmpc_free): no synchronisation
    }
    // Beaver multiplication of HOST vectors over the group (authenticated_scalar.rs:848-879 in the shape benches/batch_ops.rs:19-39 times: host
    // values in, host values out): a streaming session over the group (arkmpc_group_hostmul_*) -- every member runs its range of the same
    // vectors over its own PCIe link, the vectors are pinned once per call (not at all if the caller allocated them pinned), and the d||e
    // payload is a host vector where the network picks it up.  The triples come from the source as host vectors, or are lent in place.
    std::vector<ScalarShare> batch_mul_host(const std::vector<ScalarShare>& x, const std::vector<ScalarShare>& y) {
        if (x.size() != y.size()) throw std::invalid_argument("Cannot operate on batches of different sizes");
        std::vector<ScalarShare> out(x.size());
        batch_mul_host_into(x.data(), y.data(), x.size(), out.data());
        return out;
    }
    // the same on raw record pointers (x, y: n records each; out: n records the caller owns -- e.g. pinned memory it reuses from gate to gate)
    void batch_mul_host_into(const ScalarShare* x, const ScalarShare* y, size_t n, ScalarShare* out) {
        if (n == 0) return;
        // the triples: lent in place by a source that keeps them in memory; n copies of one record for a constant source (built once per size
        // in pinned memory and kept: offline_prep.rs:137-158 returns the same Vec every time); else the source's Vecs
        const ScalarShare *pa = nullptr, *pb = nullptr, *pc = nullptr;
        std::vector<ScalarShare> ha, hb, hc;
        ScalarShare ca, cb, cc;
        if (f_->preprocessing().borrow_triplet_batch(n, &pa, &pb, &pc)) f_->advance_ids(3 * n);
        else if (f_->preprocessing().constant_triplet(ca, cb, cc)) {
            if (const_n_ != n) {
                for (int k = 0; k < 3; ++k) {
                    if (const_trip_[k]) arkmpc_host_free(const_trip_[k]);
                    void* q = nullptr;
                    if (arkmpc_host_alloc(n * sizeof(ScalarShare), &q) != ARKMPC_OK) throw std::runtime_error("arkmpc_host_alloc failed");
                    const_trip_[k] = static_cast<ScalarShare*>(q);
                    std::fill(const_trip_[k], const_trip_[k] + n, k == 0 ? ca : (k == 1 ? cb : cc));
                }
                const_n_ = n;
            }
            pa = const_trip_[0]; pb = const_trip_[1]; pc = const_trip_[2];
            f_->advance_ids(3 * n);
        } else {
            f_->next_triple_host(n, ha, hb, hc);
            pa = ha.data(); pb = hb.data(); pc = hc.data();
        }
        std::vector<Scalar> my_de = take_scalars(2 * n);
        arkmpc_group_hostmul* s = nullptr;
        gcheck(arkmpc_group_hostmul_begin(g_, n, &x[0].share.l[0], &y[0].share.l[0], &pa[0].share.l[0], &pb[0].share.l[0], &pc[0].share.l[0], &my_de[0].l[0], &s),
               "group_hostmul_begin");
        std::vector<Scalar> peer;
        try {
            gcheck(arkmpc_group_hostmul_wait_de(s), "group_hostmul_wait_de");
            peer = f_->exchange_host_values(std::move(my_de));      // (the session's pin on my_de ended with _wait_de: the vector may move into the message)
        } catch (...) { arkmpc_group_hostmul_abort(s); throw; }
        gcheck(arkmpc_group_hostmul_finish(s, (int)f_->party_id(), f_->mac_key().l, &peer[0].l[0], &out[0].share.l[0]), "group_hostmul_finish");
        give_back(std::move(peer));
    }
    // open_authenticated_batch over the group (:278-354): opening, MAC-check shares, commit, three exchanges, verification
    GroupOpenResult open_authenticated_batch(const ShardedShares& x, const Scalar& blinder) {
        GroupOpenResult res;
        const size_t n = x.n;
        if (n == 0) return res;
        ShardedBuf mine(g_, n, 1, 4);
        gcheck(arkmpc_group_share_extract(g_, layout(), n, x.buf.cptrs(), mine.ptrs()), "group_share_extract");
        ShardedBuf peer = exchange(mine, n, 1);                                                    // round 1
        ShardedBuf opened(g_, n, 1, 4), chk(g_, n, 1, 4);
        gcheck(arkmpc_group_open_and_mac_check(g_, layout(), n, f_->mac_key().l, x.buf.cptrs(), peer.cptrs(), opened.ptrs(), chk.ptrs()), "group_open_and_mac_check");
        Scalar my_comm, recomputed;
        gcheck(arkmpc_group_commit_sha3(g_, n, chk.cptrs(), blinder.l, my_comm.l), "group_commit_sha3");
        const Scalar pc = f_->exchange_scalar(my_comm);                                            // round 2
        ShardedBuf peer_chk = exchange(chk, n, 1);                                                 // round 3
        const Scalar pb = f_->exchange_scalar(blinder);                                            // round 4
        gcheck(arkmpc_group_commit_sha3(g_, n, peer_chk.cptrs(), pb.l, recomputed.l), "group_commit_sha3(verify)");
        int ok = 0;
        gcheck(arkmpc_group_mac_verify(g_, n, chk.cptrs(), peer_chk.cptrs(), &ok), "group_mac_verify");
        res.err = (ok == 1 && std::memcmp(recomputed.l, pc.l, 32) == 0) ? MpcError::None : MpcError::AuthenticationError;
        res.value.resize(n);
        gcheck(arkmpc_group_gather_d2h(g_, n, 1, 4, opened.cptrs(), &res.value[0].l[0]), "group_gather_d2h");
        return res;
    }
    // input sharing (fabric.rs:578-600) on the single-device fabric, then sharded: the sharing step is one round and tiny next to the gates
    ShardedShares batch_share_scalar(const std::vector<Scalar>& vals_mont, size_t n, PartyId sender) {
        return shares_from_host(f_->batch_share_scalar(vals_mont, n, sender).to_host());
    }

  private:
    std::shared_ptr<MpcFabric> f_;
    arkmpc_group* g_ = nullptr;
    std::vector<std::vector<Scalar>> spare_;             // payload vectors waiting for their next exchange (take_scalars / give_back)
    ScalarShare* const_trip_[3] = {nullptr, nullptr, nullptr};   // a constant source's triple batch of const_n_ records each, pinned
    size_t const_n_ = 0;
};

// gadgets.rs:105-148 prefix_product: blind in a telescoping manner with inverse pairs, open, scan in public, unblind
inline AuthenticatedScalarBatch prefix_product(const AuthenticatedScalarBatch& values, const Scalar& blinder, MpcError* err = nullptr) {
    const size_t n = values.n;
    auto f = values.fabric;
    AuthenticatedScalarBatch b, b_inv;
    f->random_inverse_pairs(n + 1, b, b_inv);                                                             // :109
    AuthenticatedScalarBatch partial_blind = AuthenticatedScalarBatch::batch_mul(b_inv.slice(0, n), values);        // :113
    AuthenticatedScalarBatch blinded = AuthenticatedScalarBatch::batch_mul(partial_blind, b.slice(1, n));           // :114
    AuthenticatedOpenResult opened = blinded.open_authenticated_batch(blinder);                           // :117-120
    if (err) *err = opened.err;
    ScalarBatch prefixes; prefixes.n = n; prefixes.buf = DeviceBuf(f->engine(), ARKMPC_KIND_SCALAR, n);          // :131-137
    check(f->ctx(), arkmpc_scalar_prefix_product(f->ctx(), n, opened.value.buf.ptr(), prefixes.buf.ptr()), "scalar_prefix_product");
    AuthenticatedScalarBatch partial_unblind = AuthenticatedScalarBatch::batch_mul_public(b.repeat(0, n), prefixes);  // :146
    return AuthenticatedScalarBatch::batch_mul(partial_unblind, b_inv.slice(1, n));                       // :147
}

// gadgets.rs:39-52 bit_xor_batch: xor(a, b) = a + b - 2ab
inline AuthenticatedScalarBatch bit_xor_batch(const AuthenticatedScalarBatch& a, const AuthenticatedScalarBatch& b) {
    AuthenticatedScalarBatch a_plus_b = AuthenticatedScalarBatch::batch_add(a, b);
    AuthenticatedScalarBatch a_times_b = AuthenticatedScalarBatch::batch_mul(a, b);
    std::vector<Scalar> twos(a.n, a.fabric->engine()->from_u64(2));
    AuthenticatedScalarBatch t = AuthenticatedScalarBatch::batch_mul_constant(a_times_b, twos);
    return AuthenticatedScalarBatch::batch_sub(a_plus_b, t);
}

// ---------------------------------------------------------------------------------------------------------------
// lib.rs:116-128: run the same closure as both parties over an in-process duplex
// ---------------------------------------------------------------------------------------------------------------
template <class T>
std::pair<T, T> execute_mock_mpc(int field_id, int device,
                                 const std::function<std::unique_ptr<PreprocessingPhase>(PartyId, const Engine&)>& make_prep,
                                 const std::function<T(std::shared_ptr<MpcFabric>)>& f) {
    auto q01 = std::make_shared<DuplexQueue>(), q10 = std::make_shared<DuplexQueue>();
    T out[2];
    std::exception_ptr err[2];
    auto run = [&](PartyId p) {
        try {
            auto eng = std::make_shared<Engine>(field_id, device);
            std::unique_ptr<MpcNetwork> net(new MockNetwork(p, p == 0 ? q01 : q10, p == 0 ? q10 : q01));
            MpcNetwork* raw = net.get();
            auto fab = std::make_shared<MpcFabric>(p, eng, std::move(net), make_prep(p, *eng));
            if (const char* w = std::getenv("ARKMPC_MOCK_WIRE")) fab->set_wire_frames(w[0] == '1');
            if (const char* tp = std::getenv("ARKMPC_TRIPLE_PREFETCH")) fab->set_triple_prefetch(tp[0] != '0');
            if (const char* sl = std::getenv("ARKMPC_SHARE_LAYOUT")) fab->set_share_layout(std::string(sl) == "aos" ? ARKMPC_LAYOUT_AOS : ARKMPC_LAYOUT_SPLIT);
            if (const char* l = std::getenv("ARKMPC_MOCK_LINK")) {            // host | device | wire
                const std::string v = l;
                fab->set_link_mode(v == "device" ? MpcFabric::LinkMode::Device : (v == "wire" ? MpcFabric::LinkMode::Wire : MpcFabric::LinkMode::Host));
            }
            try { out[p] = f(fab); } catch (...) { raw->close(); throw; }
        } catch (...) { err[p] = std::current_exception(); }
    };
    std::thread t0(run, PARTY0), t1(run, PARTY1);
    t0.join(); t1.join();
    // report the root cause: a party that failed closes the link, which surfaces as RecvError on the other side
    auto is_recv_error = [](const std::exception_ptr& e) {
        try { std::rethrow_exception(e); } catch (const std::exception& x) { return std::string(x.what()).find("RecvError") != std::string::npos; } catch (...) { return false; }
    };
    for (auto& e : err) if (e && !is_recv_error(e)) std::rethrow_exception(e);
    for (auto& e : err) if (e) std::rethrow_exception(e);
    return {std::move(out[0]), std::move(out[1])};
}

}  // namespace arkmpc
