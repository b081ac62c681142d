// bench_main.cpp -- the reference's own criterion bench definitions, run through the host mirror (fabric.hpp) on the
// HIP engine: two parties in one process over the mock link, PartyIDBeaverSource, time = max over the two parties of the
// region the reference times.  The reference publishes no numbers for these; this driver produces
// the MI355X side of the table.
//   batch_ops        online-phase/benches/batch_ops.rs:20-39    share x, share y, batch_mul, open_authenticated_batch
//   group_batch_ops  the same for a party over several GPUs (ARKMPC_GROUP_DEVICES): batch_mul as a streaming session over the group
//   mul_throughput   benches/circuit_mul_throughput.rs:24-36    n SEQUENTIAL squarings res = res * res, then open
//   msm_throughput   benches/circuit_msm_throughput.rs:24-38    AuthenticatedPointResult::msm of n (one, identity) pairs, open
//   point_batch_mul  BASELINE config 4 secondary op: AuthenticatedPointResult::batch_mul (authenticated_curve.rs:682-714)
// usage: arkmpc_host_bench <bench> <n> [iters]      -> one JSON line
#include <chrono>
#include <cstdio>
#include <cstdlib>

#include "fabric.hpp"

using namespace arkmpc;
using Clock = std::chrono::steady_clock;

int main(int argc, char** argv) {
    if (argc < 3) { std::fprintf(stderr, "usage: %s <batch_ops|mul_throughput|msm_throughput> <n> [iters]\n", argv[0]); return 2; }
    const std::string bench = argv[1];
    const size_t n = std::strtoull(argv[2], nullptr, 10);
    const int iters = argc > 3 ? std::atoi(argv[3]) : 5;
    try {
        auto make_prep = [](PartyId p, const Engine& e) { return std::unique_ptr<PreprocessingPhase>(new PartyIDBeaverSource(p, e)); };
        auto program = [&](std::shared_ptr<MpcFabric> fabric) -> double {
            const Engine& eng = *fabric->engine();
            std::vector<Scalar> a_c(n), b_c(n);
            for (size_t i = 0; i < n; ++i) {        // any canonical values: splitmix-style words with a clear top limb
                uint64_t z = 0x9E3779B97F4A7C15ull * (i + 1);
                a_c[i] = Scalar{{z, z ^ 0xD1B54A32D192ED03ull, z * 31, z >> 8}};
                b_c[i] = Scalar{{~z, z * 7, z ^ 0x94D049BB133111EBull, z >> 9}};
            }
            std::vector<Scalar> a_m = eng.from_canonical(a_c), b_m = eng.from_canonical(b_c);
            const Scalar blinder = eng.from_u64(77 + fabric->party_id());
            double best = 1e300;
            for (int it = 0; it < iters + 1; ++it) {     // first pass warms the arena / allocations, not counted
                const auto t0 = Clock::now();
                if (bench == "batch_ops") {
                    auto a = fabric->batch_share_scalar(a_m, n, PARTY0);
                    auto b = fabric->batch_share_scalar(b_m, n, PARTY0);
                    auto res = AuthenticatedScalarBatch::batch_mul(a, b);
                    AuthenticatedOpenResult o = res.open_authenticated_batch(blinder);
                    if (o.err != MpcError::None) throw std::runtime_error("authentication failed");
                } else if (bench == "group_batch_ops") {
                    // the same shape for a party that owns several GPUs (ARKMPC_GROUP_DEVICES, default 0,0): the shares leave the sharing step as host
                    // vectors, batch_mul runs as a streaming session over the group (every member on its range of the vectors, its own link), the
                    // product is opened on the sharded path
                    std::vector<int> devs;
                    const char* gd = std::getenv("ARKMPC_GROUP_DEVICES");
                    for (std::string t = gd ? gd : "0,0"; !t.empty();) { const size_t c = t.find(','); devs.push_back(std::atoi(t.substr(0, c).c_str())); t = c == std::string::npos ? "" : t.substr(c + 1); }
                    static thread_local std::shared_ptr<GroupFabric> gf;
                    if (!gf || gf->fabric() != fabric) gf = std::make_shared<GroupFabric>(fabric, devs);
                    const bool tr = std::getenv("ARKMPC_BENCH_TRACE") && fabric->party_id() == 0;
                    auto lap = [&](const char* what) { if (tr) std::fprintf(stderr, "  %-28s %8.2f ms\n", what, std::chrono::duration<double>(Clock::now() - t0).count() * 1e3); };
                    std::vector<ScalarShare> a = fabric->batch_share_scalar(a_m, n, PARTY0).to_host();
                    std::vector<ScalarShare> b = fabric->batch_share_scalar(b_m, n, PARTY0).to_host();
                    lap("share x, y -> host vectors");
                    std::vector<ScalarShare> res = gf->batch_mul_host(a, b);
                    lap("batch_mul_host");
                    ShardedShares sres = gf->shares_from_host(res);
                    lap("shares_from_host");
                    GroupOpenResult o = gf->open_authenticated_batch(sres, blinder);
                    lap("open_authenticated_batch");
                    if (o.err != MpcError::None) throw std::runtime_error("authentication failed");
                } else if (bench == "mul_throughput") {
                    auto res = fabric->batch_share_scalar(std::vector<Scalar>(1, eng.from_u64(1)), 1, PARTY0);
                    for (size_t k = 0; k < n; ++k) res = AuthenticatedScalarBatch::batch_mul(res, res);
                    ScalarBatch o = res.open_batch();
                    (void)o.to_host();
                } else if (bench == "msm_throughput") {
                    // one_authenticated = (party_id, mac_key share) (fabric.rs:226-252); curve_identity_authenticated = identity shares
                    std::vector<ScalarShare> ones(n, ScalarShare{eng.from_u64(fabric->party_id()), fabric->mac_key()});
                    AuthenticatedScalarBatch scalars = fabric->allocate_scalar_shares(ones);
                    AuthenticatedPointBatch points = AuthenticatedPointBatch::alloc(fabric, n);
                    {
                        std::vector<uint64_t> id(24 * n, 0);
                        const Scalar one_q = Engine(3, 0).from_u64(1);          // Fq Montgomery one: identity = (1, 1, 0)
                        for (size_t i = 0; i < 2 * n; ++i) { std::memcpy(&id[12 * i], one_q.l, 32); std::memcpy(&id[12 * i + 4], one_q.l, 32); }
                        points.buf.upload(id.data(), n * 192);
                    }
                    const auto t1 = Clock::now();
                    auto res = AuthenticatedPointBatch::msm(scalars, points);
                    PointBatch o = res.open_batch();
                    (void)o.to_host();
                    const double s = std::chrono::duration<double>(Clock::now() - t1).count();
                    if (it) best = s < best ? s : best;
                    continue;
                } else if (bench == "point_batch_mul") {
                    // BASELINE config 4, secondary op: AuthenticatedPointResult::batch_mul (authenticated_curve.rs:682-714),
                    // [x * yG] by a Beaver triple: the reference's 10 scalar-muls per element and party regrouped to 6 (fabric.hpp batch_mul;
                    // ARKMPC_POINT_MUL_LITERAL=1 runs the literal sequence); timed without the sharing
                    auto x = fabric->batch_share_scalar(a_m, n, PARTY0);
                    auto y = fabric->batch_share_scalar(b_m, n, PARTY1);
                    auto Y = AuthenticatedPointBatch::batch_mul_generator(y);
                    check(fabric->ctx(), arkmpc_sync(fabric->ctx()), "sync");
                    const auto t1 = Clock::now();
                    auto Z = AuthenticatedPointBatch::batch_mul(x, Y);
                    check(fabric->ctx(), arkmpc_sync(fabric->ctx()), "sync");
                    const double s = std::chrono::duration<double>(Clock::now() - t1).count();
                    if (it) best = s < best ? s : best;
                    continue;
                } else {
                    throw std::invalid_argument("unknown bench " + bench);
                }
                const double s = std::chrono::duration<double>(Clock::now() - t0).count();
                if (it) best = s < best ? s : best;
            }
            return best;
        };
        auto both = execute_mock_mpc<double>(0, 0, make_prep, program);
        const double t = both.first > both.second ? both.first : both.second;
        std::printf("{\"bench\": \"%s\", \"n\": %zu, \"iters\": %d, \"link\": \"%s\", \"seconds\": %.6g, \"elements_per_s\": %.6g}\n", bench.c_str(), n, iters,
                    std::getenv("ARKMPC_MOCK_LINK") ? std::getenv("ARKMPC_MOCK_LINK") : "host", t, (double)n / t);
    } catch (const std::exception& e) {
        std::fprintf(stderr, "host bench failed: %s\n", e.what());
        return 1;
    }
    return 0;
}
