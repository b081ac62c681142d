// mock_mpc_main.cpp -- two-party scenarios over the in-process mock network, written against fabric.hpp the way the
// reference's own tests are written against ark-mpc (execute_mock_mpc + PartyIDBeaverSource, lib.rs:116-128):
//   batch_mul       benches/batch_ops.rs:20-39 / authenticated_scalar.rs test_batch_mul (:1571-1594)
//   share_and_open  integration/src/fabric.rs:15-32
//   circuit         add / sub / neg / mul / mul_public / add_public / sub_public composition
//   + --bad-mac / --bad-share: integration/src/authenticated_scalar.rs:49-75 (open_authenticated must fail)
// usage: arkmpc_mock_mpc <scenario> <field_id> <n> <in_file> <out_file> [--bad-mac|--bad-share]
// in_file : n canonical a values then n canonical b values (32-byte little-endian each)
// out_file: per party: u64 error code (0 ok, 2 AuthenticationError), then n canonical opened values
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>

#include "fabric.hpp"

using namespace arkmpc;

struct PartyOut {
    uint64_t err = 0;
    std::vector<Scalar> opened;   // canonical
};

// Point-side scenarios, generic over the curve like the reference's tests (BN254 G1 for field 0, Curve25519 for field 2).
// Output: number of failed per-element MAC checks, then the opened points in the compressed encoding (32 bytes each).
template <class APB>
static PartyOut point_scenario(std::shared_ptr<MpcFabric> fabric, const std::string& scenario, size_t n, const std::vector<Scalar>& a_m,
                               const std::vector<Scalar>& b_m, bool bad_mac) {
    using Cv = typename APB::Curve;
    using PB = typename APB::PointBatch;
    const Engine& eng = *fabric->engine();
    auto gen_mul = [&](const std::vector<Scalar>& s_m) {          // [s_i]G as a public batch
        ScalarBatch sc = fabric->allocate_scalars(s_m);
        PB pg = APB::alloc_points(fabric, s_m.size());
        if (!s_m.empty()) check(fabric->ctx(), Cv::generator_mul(fabric->ctx(), s_m.size(), sc.buf.ptr(), pg.buf.ptr()), "generator_mul");
        return pg;
    };
    APB Z;
    uint64_t forms_mismatch = 0;
    if (scenario == "share_point") {
        // batch_share_point (fabric.rs:622-649): party 0 shares P_i = [a_i]G; open_authenticated must return P_i
        std::vector<uint64_t> pts = gen_mul(a_m).to_host();       // both parties can compute it here; only the sender's copy is used
        Z = fabric->batch_share_point<APB>(pts, n, PARTY0);
    } else if (scenario == "point_sub_public") {
        // AuthenticatedPointResult - CurvePointResult (authenticated_curve.rs:553-575 -> PointShare::sub_public, curve/share.rs:63-65):
        // share [a_i]G, subtract the public [b_i]G: opens to [(a_i - b_i)]G with a valid MAC
        std::vector<uint64_t> pts = gen_mul(a_m).to_host();
        APB A = fabric->batch_share_point<APB>(pts, n, PARTY1);
        Z = APB::batch_sub_public(A, gen_mul(b_m));
    } else {
        // AuthenticatedPointResult::batch_mul (authenticated_curve.rs:682-714, test :1222-1244): share x, share y,
        // Y = [y]G, Z = batch_mul(x, Y), open_authenticated; out = compressed points of Z  (expected (x*y) G)
        auto x = fabric->batch_share_scalar(a_m, n, PARTY0);
        auto y = fabric->batch_share_scalar(b_m, n, PARTY1);
        auto Y = APB::batch_mul_generator(y);
        if (scenario == "msm_public_points") Z = APB::msm_authenticated(x, gen_mul(b_m));     // curve.rs:618-642: bases [b_i]G public
        else if (scenario == "point_mul_forms") {
            // the regrouped form the engine runs, ([a] + d) eG + ([c] + d[b]) G, against the reference's literal op sequence
            // (authenticated_curve.rs:696-713): EVERY local share and MAC point must be the same group element
            AuthenticatedScalarBatch ta, tb, tc;
            fabric->next_triple_batch(n, ta, tb, tc);                 // ONE triple for both forms: with it, the local shares must coincide
            Z = APB::batch_mul_with_triple(x, Y, ta, tb, tc, false);
            APB Zl = APB::batch_mul_with_triple(x, Y, ta, tb, tc, true);
            std::vector<uint8_t> b1(64 * n + 8), b2(64 * n + 8);
            DeviceBuf d1(fabric->engine(), 8 * n + 1), d2(fabric->engine(), 8 * n + 1);
            check(fabric->ctx(), Cv::to_bytes(fabric->ctx(), 2 * n, Z.buf.ptr(), reinterpret_cast<uint8_t*>(d1.ptr())), "to_bytes");
            check(fabric->ctx(), Cv::to_bytes(fabric->ctx(), 2 * n, Zl.buf.ptr(), reinterpret_cast<uint8_t*>(d2.ptr())), "to_bytes");
            d1.download(b1.data(), 64 * n); d2.download(b2.data(), 64 * n);
            for (size_t i = 0; i < 2 * n; ++i) if (std::memcmp(&b1[32 * i], &b2[32 * i], 32) != 0) forms_mismatch += 1;
        }
        else Z = (scenario == "msm") ? APB::msm(x, Y) : APB::batch_mul(x, Y);
    }
    const size_t zn = Z.n;    // msm collapses the batch to one point
    if (fabric->party_id() == PARTY0 && zn && bad_mac) {          // corrupt one MAC point: make it the share point
        std::vector<uint64_t> h(2 * Cv::PW * zn); Z.buf.download(h.data(), h.size() * 8);
        std::memcpy(&h[2 * Cv::PW * (zn / 2) + Cv::PW], &h[2 * Cv::PW * (zn / 2)], Cv::PW * 8);
        Z.buf.upload(h.data(), h.size() * 8);
    }
    std::vector<Scalar> bl(zn);
    for (size_t i = 0; i < zn; ++i) bl[i] = eng.from_u64(1000 + 7 * i + fabric->party_id());
    auto o = Z.open_authenticated_batch(bl);
    PartyOut out;
    for (size_t i = 0; i < zn; ++i) if (!o.ok[i]) out.err += 1;          // number of failed MAC checks
    out.err += 1000000 * forms_mismatch;                                 // point_mul_forms: local shares that differ between the two forms
    out.opened.resize(zn);
    if (zn) {
        DeviceBuf bytes(fabric->engine(), 4 * zn);
        check(fabric->ctx(), Cv::to_bytes(fabric->ctx(), zn, o.value.buf.ptr(), reinterpret_cast<uint8_t*>(bytes.ptr())), "to_bytes");
        bytes.download(out.opened.data(), zn * 32);
    }
    return out;
}

int main(int argc, char** argv) {
    if (argc < 6) { std::fprintf(stderr, "usage: %s <scenario> <field_id> <n> <in_file> <out_file> [--bad-mac|--bad-share]\n", argv[0]); return 2; }
    const std::string scenario = argv[1];
    const int field_id = std::atoi(argv[2]);
    const size_t n = std::strtoull(argv[3], nullptr, 10);
    bool bad_mac = false, bad_share = false;
    for (int i = 6; i < argc; ++i) { if (std::string(argv[i]) == "--bad-mac") bad_mac = true; if (std::string(argv[i]) == "--bad-share") bad_share = true; }
    std::vector<Scalar> a_c(n), b_c(n);
    {
        std::ifstream in(argv[4], std::ios::binary);
        in.read(reinterpret_cast<char*>(a_c.data()), n * 32);
        in.read(reinterpret_cast<char*>(b_c.data()), n * 32);
        if (!in) { std::fprintf(stderr, "short input file\n"); return 2; }
    }
    try {
        // ARKMPC_MOCK_DEALER=<seed>: a trusted-dealer source with random MAC key shares and triples instead of the reference's constant dummy source
        const char* dealer = std::getenv("ARKMPC_MOCK_DEALER");
        const uint64_t dealer_seed = dealer ? std::strtoull(dealer, nullptr, 0) : 0;
        // ARKMPC_MOCK_VECTOR_TRIPLES=<count>: the triples are drawn from that source up front into pinned host memory and lent to the fabric in
        // place, gate after gate (VectorBeaverSource: what a real offline phase's output looks like to this engine)
        const char* vec = std::getenv("ARKMPC_MOCK_VECTOR_TRIPLES");
        const size_t vec_cap = vec ? std::strtoull(vec, nullptr, 0) : 0;
        auto make_prep = [&](PartyId p, const Engine& e) {
            std::unique_ptr<PreprocessingPhase> src = dealer ? std::unique_ptr<PreprocessingPhase>(new DealerBeaverSource(p, e, dealer_seed))
                                                             : std::unique_ptr<PreprocessingPhase>(new PartyIDBeaverSource(p, e));
            if (vec) return std::unique_ptr<PreprocessingPhase>(new VectorBeaverSource(std::move(src), vec_cap));
            return src;
        };
        auto program = [&](std::shared_ptr<MpcFabric> fabric) -> PartyOut {
            const Engine& eng = *fabric->engine();
            std::vector<Scalar> a_m = eng.from_canonical(a_c), b_m = eng.from_canonical(b_c);
            const Scalar blinder = eng.from_u64(0x1234567 + fabric->party_id());
            AuthenticatedScalarBatch res;
            if (scenario == "batch_mul") {
                auto a = fabric->batch_share_scalar(a_m, n, PARTY0);
                auto b = fabric->batch_share_scalar(b_m, n, PARTY0);
                res = AuthenticatedScalarBatch::batch_mul(a, b);
            } else if (scenario == "share_and_open") {
                res = fabric->batch_share_scalar(a_m, n, PARTY1);
            } else if (scenario == "circuit") {
                auto a = fabric->batch_share_scalar(a_m, n, PARTY0);
                auto b = fabric->batch_share_scalar(b_m, n, PARTY1);
                auto t = AuthenticatedScalarBatch::batch_add(a, b);
                auto u = AuthenticatedScalarBatch::batch_sub(a, b);
                auto v = AuthenticatedScalarBatch::batch_mul(t, u);                       // a^2 - b^2
                auto w = AuthenticatedScalarBatch::batch_neg(v);
                ScalarBatch pa = fabric->batch_share_plaintext(a_m, n, PARTY0);           // a as a public value
                ScalarBatch pb = b.open_batch();                                          // b opened (unauthenticated)
                auto x = AuthenticatedScalarBatch::batch_mul_public(w, pa);
                auto y = AuthenticatedScalarBatch::batch_add_public(x, pa);
                res = AuthenticatedScalarBatch::batch_sub_public(y, pb);                  // -(a^2-b^2)*a + a - b
            } else if (scenario == "prefix_product") {
                // gadgets.rs:105-148 / test_prefix_product: open(prefix_product(x))_i == x_0 * ... * x_i
                auto a = fabric->batch_share_scalar(a_m, n, PARTY0);
                res = prefix_product(a, eng.from_u64(999 + fabric->party_id()));
            } else if (scenario == "xor") {
                // gadgets.rs bit_xor_batch / authenticated_scalar.rs test_xor_circuit (:1677-1688): a + b - 2ab on shared bits
                auto a = fabric->batch_share_scalar(a_m, n, PARTY0);
                auto b = fabric->batch_share_scalar(b_m, n, PARTY1);
                res = bit_xor_batch(a, b);
            } else if (scenario == "div") {
                // batch_div (authenticated_scalar.rs:974-977): open(a / b) == a * b^-1
                auto a = fabric->batch_share_scalar(a_m, n, PARTY0);
                auto b = fabric->batch_share_scalar(b_m, n, PARTY1);
                res = AuthenticatedScalarBatch::batch_div(a, b, eng.from_u64(555 + fabric->party_id()));
            } else if (scenario == "inverse") {
                // AuthenticatedScalarResult::batch_inverse (authenticated_scalar.rs:55-82, test :1640-1660): open(inverse(x)) == x^-1
                auto a = fabric->batch_share_scalar(a_m, n, PARTY0);
                res = AuthenticatedScalarBatch::batch_inverse(a, eng.from_u64(777 + fabric->party_id()));
            } else if (scenario == "tail") {
                // the small API tail in one circuit: pow (authenticated_scalar.rs:86-100, incl. pow(0) = the shared ZERO wire and pow(1) = clone),
                // batch_add_constant (:531-560), Sum for AuthenticatedScalarResult (:563-575), ones_authenticated (fabric.rs:525-534),
                // Sum / Product for ScalarResult (scalar_result.rs:325-338), ScalarResult::batch_add_constant / batch_sub_constant (:119, :205)
                //   res_i = a_i^5 + b_i + S + 0 + 1 + P + T - a_i^1 ... with S = sum_j (a_j^5 + b_j), P = prod_j b_j, T = sum_j b_j
                auto a = fabric->batch_share_scalar(a_m, n, PARTY0);
                auto q = AuthenticatedScalarBatch::batch_add_constant(AuthenticatedScalarBatch::pow(a, 5), b_m);
                auto S = AuthenticatedScalarBatch::sum(q).repeat(0, n);
                auto z = AuthenticatedScalarBatch::pow(a, 0);
                auto one = fabric->ones_authenticated(n);
                ScalarBatch pb = fabric->batch_share_plaintext(b_m, n, PARTY1);
                const Scalar P = scalar_batch_reduce(fabric->engine(), pb, true).to_host()[0], T = scalar_batch_reduce(fabric->engine(), pb, false).to_host()[0];
                ScalarBatch Pn = fabric->allocate_scalars(std::vector<Scalar>(n, P)), Tn = fabric->allocate_scalars(std::vector<Scalar>(n, T));
                ScalarBatch PT = scalar_batch_addsub(fabric->engine(), scalar_batch_addsub(fabric->engine(), Pn, Tn, false), pb, true);       // P + T - b_i
                auto r1 = AuthenticatedScalarBatch::batch_add(AuthenticatedScalarBatch::batch_add(q, S), AuthenticatedScalarBatch::batch_add(z, one));
                res = AuthenticatedScalarBatch::batch_sub(AuthenticatedScalarBatch::batch_add_public(r1, PT), AuthenticatedScalarBatch::pow(a, 1));
            } else if (scenario == "group_mul") {
                // one party over several GPUs: the same circuit as "batch_mul" + the authenticated opening, range-sharded over the members of
                // an arkmpc_group (ARKMPC_GROUP_DEVICES = comma-separated device ids, repeats allowed; default 0,0,0)
                std::vector<int> devs;
                const char* gd = std::getenv("ARKMPC_GROUP_DEVICES");
                for (std::string t = gd ? gd : "0,0,0"; !t.empty();) { const size_t c = t.find(','); devs.push_back(std::atoi(t.substr(0, c).c_str())); t = c == std::string::npos ? "" : t.substr(c + 1); }
                auto gf = std::make_shared<GroupFabric>(fabric, devs);
                auto a = gf->batch_share_scalar(a_m, n, PARTY0);
                auto b = gf->batch_share_scalar(b_m, n, PARTY1);
                auto prod = gf->batch_mul(a, b);
                auto sq = gf->batch_mul(prod, prod);                                   // a second, dependent gate on sharded operands
                if (fabric->party_id() == PARTY0 && n && (bad_mac || bad_share)) {
                    std::vector<ScalarShare> h = gf->to_host(sq);
                    (bad_mac ? h[n - 1].mac : h[n - 1].share) = eng.from_u64(42);       // the LAST element: the last member's range
                    sq = gf->shares_from_host(h);
                }
                GroupOpenResult o = gf->open_authenticated_batch(sq, blinder);
                PartyOut out;
                out.err = (o.err == MpcError::None) ? 0 : 2;
                if (n) out.opened = eng.to_canonical(o.value);
                return out;
            } else if (scenario == "chain") {
                // a depth-4 chain of dependent gates on resident operands, z <- z * b, then a gate of ANOTHER size on a slice (served from the
                // read-ahead triples, in order) and one more of the full size: every gate pulls fresh triples from the source (read ahead by one
                // gate unless ARKMPC_TRIPLE_PREFETCH=0); opens to 2 a b^5 (tests/test_host_fabric.py)
                auto z = fabric->batch_share_scalar(a_m, n, PARTY0);
                auto b = fabric->batch_share_scalar(b_m, n, PARTY1);
                for (int k = 0; k < 4; ++k) z = AuthenticatedScalarBatch::batch_mul(z, b);
                const size_t h = n / 2;
                auto zh = AuthenticatedScalarBatch::batch_mul(z.slice(0, h), b.slice(0, h));             // a smaller request
                auto zf = AuthenticatedScalarBatch::batch_mul(z, b);                                     // a larger one than what is left of the read-ahead
                res = AuthenticatedScalarBatch::batch_add(zf, zf);
                if (h) {
                    auto fix = AuthenticatedScalarBatch::batch_sub(zh, zf.slice(0, h));                  // zh == zf on the first half: adds zero there
                    std::vector<ScalarShare> add = fix.to_host(), all(n, ScalarShare{Scalar{{0, 0, 0, 0}}, Scalar{{0, 0, 0, 0}}});
                    std::copy(add.begin(), add.end(), all.begin());
                    res = AuthenticatedScalarBatch::batch_add(res, fabric->allocate_scalar_shares(all));
                }
            } else if (scenario == "group_mul_host") {
                // GroupFabric::batch_mul_host: host records in, host records out, a streaming session over the group's members; then the
                // authenticated opening of the product on the sharded path
                std::vector<int> devs;
                const char* gd = std::getenv("ARKMPC_GROUP_DEVICES");
                for (std::string t = gd ? gd : "0,0,0"; !t.empty();) { const size_t c = t.find(','); devs.push_back(std::atoi(t.substr(0, c).c_str())); t = c == std::string::npos ? "" : t.substr(c + 1); }
                std::vector<ScalarShare> ha = fabric->batch_share_scalar(a_m, n, PARTY0).to_host(), hb = fabric->batch_share_scalar(b_m, n, PARTY1).to_host();
                auto gf = std::make_shared<GroupFabric>(fabric, devs);
                std::vector<ScalarShare> prod = gf->batch_mul_host(ha, hb);
                std::vector<ScalarShare> sq = gf->batch_mul_host(prod, prod);
                if (fabric->party_id() == PARTY0 && n && (bad_mac || bad_share)) (bad_mac ? sq[n - 1].mac : sq[n - 1].share) = eng.from_u64(42);
                GroupOpenResult o = gf->open_authenticated_batch(gf->shares_from_host(sq), blinder);
                PartyOut out;
                out.err = (o.err == MpcError::None) ? 0 : 2;
                if (n) out.opened = eng.to_canonical(o.value);
                return out;
            } else if (scenario == "short_peer") {
                // a peer that sends one element fewer than the protocol step requires: must surface as a network error on the
                // honest side before any kernel reads the short buffer (never an out-of-bounds read)
                auto a = fabric->batch_share_scalar(a_m, n, PARTY0);
                res = (fabric->party_id() == PARTY1 && n) ? a.slice(0, n - 1) : std::move(a);
            } else if (scenario == "shared_bits") {
                // fabric.rs:961-984 random_shared_bits over PreprocessingPhase::next_shared_bit_batch (offline_prep.rs:39-44): the
                // dummy source's "bit" is the party id, so the authenticated open gives 0 + 1 = 1 for every element
                res = fabric->random_shared_bits(n);
            } else if (scenario == "share_point" || scenario == "point_mul" || scenario == "point_mul_forms" || scenario == "msm" || scenario == "msm_public_points" ||
                       scenario == "point_sub_public") {
                return field_id == ARKMPC_CURVE25519_FR ? point_scenario<AuthenticatedEdPointBatch>(fabric, scenario, n, a_m, b_m, bad_mac)
                                                        : point_scenario<AuthenticatedPointBatch>(fabric, scenario, n, a_m, b_m, bad_mac);
            } else {
                throw std::invalid_argument("unknown scenario " + scenario);
            }
            if (fabric->party_id() == PARTY0 && n) {
                if (bad_mac) res.modify_mac(n / 2, eng.from_u64(42));
                if (bad_share) res.modify_share(n / 2, eng.from_u64(42));
            }
            AuthenticatedOpenResult o = res.open_authenticated_batch(blinder);
            PartyOut out;
            out.err = (o.err == MpcError::None) ? 0 : 2;
            if (n) out.opened = eng.to_canonical(o.value.to_host());
            return out;
        };
        auto both = execute_mock_mpc<PartyOut>(field_id, 0, make_prep, program);
        std::printf("frames %llu\n", (unsigned long long)MpcFabric::frames_sent().load());
        std::ofstream out(argv[5], std::ios::binary);
        for (const PartyOut* po : {&both.first, &both.second}) {
            out.write(reinterpret_cast<const char*>(&po->err), 8);
            out.write(reinterpret_cast<const char*>(po->opened.data()), po->opened.size() * 32);
        }
    } catch (const std::exception& e) {
        std::fprintf(stderr, "mock mpc failed: %s\n", e.what());
        return 1;
    }
    return 0;
}
