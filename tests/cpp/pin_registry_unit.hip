// pin_registry_unit.hip -- host-only unit test of the bookkeeping behind "zero-copy kernels only see addresses in their first registered life"
// (csrc/arkmpc_internal.hpp PinRegistry: the retired-interval set and the entry lookups).  No HIP call is made: runs without a GPU.
// Built and run by tests/test_abi_cpu.py with hipcc (the header pulls in hip_runtime.h).
#include <cstdio>
#include <random>
#include <set>

#include "../../ark-mpc_amd/csrc/arkmpc_internal.hpp"

#define CHECK(c) do { if (!(c)) { std::printf("FAILED line %d: %s\n", __LINE__, #c); return 1; } } while (0)

int main() {
    const uintptr_t P = PinRegistry::kPage;
    {   // retire rounds out to pages, merges touching and overlapping intervals, answers overlap queries
        PinRegistry r;
        r.retire(10 * P + 5, 100);                                   // -> [10P, 11P)
        CHECK(r.retired.size() == 1 && r.retired.begin()->first == 10 * P && r.retired.begin()->second == 11 * P);
        CHECK(r.retired_overlaps(10 * P, 1) && r.retired_overlaps(11 * P - 1, 1) && !r.retired_overlaps(11 * P, 1) && !r.retired_overlaps(9 * P, P));
        CHECK(r.retired_overlaps(9 * P, P + 1));                     // one byte into the retired page
        r.retire(20 * P, 3 * P);                                     // a second interval
        r.retire(11 * P, P);                                         // touches the first: merged
        CHECK(r.retired.size() == 2 && r.retired.begin()->second == 12 * P);
        r.retire(11 * P + 1, 9 * P);                                 // bridges both
        CHECK(r.retired.size() == 1 && r.retired.begin()->first == 10 * P && r.retired.begin()->second == 23 * P);
        r.retire(0, P);                                              // before everything
        CHECK(r.retired.size() == 2 && r.retired_overlaps(0, 1) && !r.retired_overlaps(P, 9 * P) && r.retired_overlaps(P, 9 * P + 1));
        r.retire(5 * P, 40 * P);                                     // swallows the big one
        CHECK(r.retired.size() == 2 && r.retired.rbegin()->first == 5 * P && r.retired.rbegin()->second == 45 * P);
    }
    {   // against a page bitmap, random ranges
        PinRegistry r;
        std::mt19937_64 g(7);
        std::set<uintptr_t> pages;
        for (int it = 0; it < 3000; ++it) {
            const uintptr_t lo = (g() % 4000) * P + g() % P, len = 1 + g() % (6 * P);
            if (it % 3) {
                r.retire(lo, len);
                for (uintptr_t p = lo / P; p <= (lo + len - 1) / P; ++p) pages.insert(p);
            } else {
                bool want = false;
                for (uintptr_t p = lo / P; p <= (lo + len - 1) / P; ++p) want = want || pages.count(p);
                CHECK(r.retired_overlaps(lo, len) == want);
            }
        }
        uintptr_t prev_hi = 0;
        for (auto& iv : r.retired) { CHECK(iv.first < iv.second && (prev_hi == 0 || iv.first > prev_hi)); prev_hi = iv.second; }     // disjoint, not touching, sorted
    }
    {   // the set is bounded: beyond kMaxRetired intervals the smallest gaps are closed -- it only ever grows (safe side)
        PinRegistry r;
        for (uintptr_t k = 0; k < PinRegistry::kMaxRetired + 500; ++k) r.retire((3 * k + 1) * P, P);
        CHECK(r.retired.size() <= PinRegistry::kMaxRetired);
        for (uintptr_t k = 0; k < PinRegistry::kMaxRetired + 500; ++k) CHECK(r.retired_overlaps((3 * k + 1) * P, 1));               // nothing retired was forgotten
    }
    {   // entries: containment and overlap (no runtime calls: the entries are put in by hand)
        PinRegistry r;
        r.ents[100 * P] = PinRegistry::Ent{4 * P, 1, 0, false, false};
        r.ents[200 * P] = PinRegistry::Ent{P, 1, 1, true, false};
        CHECK(r.containing(100 * P, 4 * P) != r.ents.end() && r.containing(101 * P, P) != r.ents.end());
        CHECK(r.containing(100 * P, 4 * P + 1) == r.ents.end() && r.containing(99 * P, 2 * P) == r.ents.end() && r.containing(50 * P, P) == r.ents.end());
        CHECK(r.overlaps_entry(99 * P, P + 1) && r.overlaps_entry(103 * P, 10 * P) && !r.overlaps_entry(104 * P, P) && !r.overlaps_entry(0, 100 * P));
        CHECK(r.overlaps_entry(150 * P, 51 * P) && !r.overlaps_entry(150 * P, 50 * P));
        bool ours = false, reused = false;
        CHECK(r.lookup((void*)(100 * P + 8), 64, &ours, &reused) && ours && !reused);
        CHECK(r.lookup((void*)(200 * P), P, &ours, &reused) && !ours && reused);
        CHECK(!r.lookup((void*)(103 * P), 2 * P, &ours, &reused));
    }
    std::printf("pin registry unit ok\n");
    return 0;
}
