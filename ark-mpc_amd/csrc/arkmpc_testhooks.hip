// arkmpc_testhooks.hip -- test-only entry points that must NOT ship in libarkmpc_hip.so (include/arkmpc_test_hooks.h).  Built into
// ark-mpc_amd/lib/libarkmpc_testhooks.so; only tests/ load it.
#include "arkmpc_internal.hpp"

namespace {
int throwing_body(arkmpc_ctx* ctx) {                       // stands for any entry point: the guard every entry point holds, then a body that throws
    if (!ctx) return ARKMPC_ERR_BAD_ARG;
    CtxGuard guard(ctx);
    if (guard.rc) return guard.rc;
    std::vector<int> v;
    if (ctx->device >= 0) throw std::bad_alloc();          // (a condition the optimiser cannot fold: the return below stays reachable)
    return (int)v.size();
}
}  // namespace

extern "C" int arkmpc_test_throw_inside(arkmpc_ctx* ctx) {
    // the catch stands for a caller with a handler or landing pad of its own (a Rust frame): with one in sight the runtime unwinds through the
    // entry point, and the guard must end the process before control gets here.  (With no handler anywhere the runtime terminates without
    // unwinding at all -- the same abort, by another road.)
    try { return throwing_body(ctx); } catch (...) { return -99; }
}
