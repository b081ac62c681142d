// prefetch_fault.cpp -- what happens to triples the fabric has CONSUMED from its source when their upload fails, and what the read-ahead
// consumes at the tail of a circuit (round-5 advisor finding on host/fabric.hpp prefetch_triples).  Test infrastructure: built and run by
// tests/test_host_fabric.py.  The fault is injected by interposition: this program DEFINES arkmpc_batch_from_host_async, so the calls the
// header-only host mirror makes bind here; the real entry point is reached through dlsym(RTLD_NEXT).  No product code knows about it.
//
//   prefetch_fault <n> <gates> <capacity> <fail_call> <last_gate_hint>
//     a depth-<gates> chain z <- z * b on n elements, both parties in-process, triples from a VectorBeaverSource (over a dealer source) holding
//     <capacity> triples; the <fail_call>-th call of arkmpc_batch_from_host_async in the PROCESS (0-based, both parties count; -1 = none) returns
//     ARKMPC_ERR_HIP without doing anything; last_gate_hint = 1: set_triple_prefetch(false) before the last gate.
//   prints one line: ok <consumed by party 0> <consumed by party 1> <xor-checksum of the opened values> <async calls seen>
#include <dlfcn.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>

#include "fabric.hpp"

using namespace arkmpc;

static std::atomic<long> g_calls{0};
static long g_fail_call = -1;

extern "C" int arkmpc_batch_from_host_async(arkmpc_ctx* ctx, int kind, int layout, size_t n, const void* host_records, arkmpc_batch** out_batch) {
    using Fn = int (*)(arkmpc_ctx*, int, int, size_t, const void*, arkmpc_batch**);
    static Fn real = reinterpret_cast<Fn>(dlsym(RTLD_NEXT, "arkmpc_batch_from_host_async"));
    const long k = g_calls.fetch_add(1);
    if (k == g_fail_call) { if (out_batch) *out_batch = nullptr; return ARKMPC_ERR_HIP; }
    return real(ctx, kind, layout, n, host_records, out_batch);
}

struct Out {
    uint64_t err = 0, consumed = 0, sum = 0;
};

int main(int argc, char** argv) {
    if (argc < 6) { std::fprintf(stderr, "usage: %s <n> <gates> <capacity> <fail_call> <last_gate_hint>\n", argv[0]); return 2; }
    const size_t n = std::strtoull(argv[1], nullptr, 10), gates = std::strtoull(argv[2], nullptr, 10), cap = std::strtoull(argv[3], nullptr, 10);
    g_fail_call = std::atol(argv[4]);
    const bool hint = std::atoi(argv[5]) != 0;
    try {
        VectorBeaverSource* src[2] = {nullptr, nullptr};
        auto make_prep = [&](PartyId p, const Engine& e) {
            auto* v = new VectorBeaverSource(std::unique_ptr<PreprocessingPhase>(new DealerBeaverSource(p, e, 0xFA17)), cap);
            src[p] = v;
            return std::unique_ptr<PreprocessingPhase>(v);
        };
        std::function<Out(std::shared_ptr<MpcFabric>)> program = [&](std::shared_ptr<MpcFabric> fabric) {
            const Engine& eng = *fabric->engine();
            std::vector<Scalar> a(n), b(n);
            for (size_t i = 0; i < n; ++i) { a[i] = eng.from_u64(3 + i); b[i] = eng.from_u64(5 + 2 * i); }
            auto z = fabric->batch_share_scalar(a, n, PARTY0);
            auto y = fabric->batch_share_scalar(b, n, PARTY1);
            for (size_t g = 0; g < gates; ++g) {
                if (hint && g + 1 == gates) fabric->set_triple_prefetch(false);
                z = AuthenticatedScalarBatch::batch_mul(z, y);
            }
            AuthenticatedOpenResult o = z.open_authenticated_batch(eng.from_u64(77 + fabric->party_id()));
            Out out;
            out.err = (o.err == MpcError::None) ? 0 : 2;
            for (const Scalar& s : eng.to_canonical(o.value.to_host())) for (int k = 0; k < 4; ++k) out.sum ^= s.l[k] * 0x9E3779B97F4A7C15ull + (out.sum << 7);
            out.consumed = cap - src[fabric->party_id()]->remaining();
            return out;
        };
        auto both = execute_mock_mpc<Out>(ARKMPC_BN254_FR, 0, make_prep, program);
        if (both.first.err || both.second.err) { std::printf("mac_check_failed %llu %llu\n", (unsigned long long)both.first.err, (unsigned long long)both.second.err); return 1; }
        if (both.first.sum != both.second.sum) { std::printf("parties_disagree\n"); return 1; }
        std::printf("ok %llu %llu %016llx %ld\n", (unsigned long long)both.first.consumed, (unsigned long long)both.second.consumed, (unsigned long long)both.first.sum, g_calls.load());
    } catch (const std::exception& e) {
        std::fprintf(stderr, "prefetch_fault failed: %s\n", e.what());
        return 1;
    }
    return 0;
}
