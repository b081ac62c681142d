/*
 * arkmpc_test_hooks.h -- entry points that exist for the library's OWN tests.  Not part of the drop-in boundary (include/arkmpc.h): nothing a
 * caller of the reference's API needs, no stability promise, absent from the generated Rust FFI.  Declared here so that every exported
 * `arkmpc_*` symbol is declared in exactly one of the two headers (tests/test_abi_cpu.py checks the export tables against both).
 * libarkmpc_hip.so exports only the arithmetic self-test below (it computes, nothing else); the hook that ends the process lives in a
 * test-only library of its own.  (Which path a streaming session took used to be a process-wide test counter here: it is now part of the
 * boundary's per-context diagnostics, arkmpc_ctx_get_stats.)
 */
#ifndef ARKMPC_TEST_HOOKS_H
#define ARKMPC_TEST_HOOKS_H
#include "arkmpc.h"
#ifdef __cplusplus
extern "C" {
#endif
/* the nine-29-bit-limb plain arithmetic of the Curve25519 MSM kernels (csrc/arkmpc_edwards.hip k_f9_selftest) on n pairs of 256-bit values:
 * out = n x 5 results of 4 x u64 each: a*b, a+b, a-b, (a-b)*(a+b), 1/a mod 2^255 - 19 */
int arkmpc_test_f9(arkmpc_ctx* ctx, size_t n, const uint64_t* a, const uint64_t* b, uint64_t* out);
/* throws std::bad_alloc from inside an entry point's body: the process must end with the library's message on stderr (abort), never unwind
 * into the caller (csrc/arkmpc_internal.hpp CtxGuard).  Run it in a child process.  NOT in libarkmpc_hip.so: a symbol that aborts the process
 * by design has no place in the library a caller links -- it is built into ark-mpc_amd/lib/libarkmpc_testhooks.so (csrc/arkmpc_testhooks.hip),
 * which only the tests load. */
int arkmpc_test_throw_inside(arkmpc_ctx* ctx);
#ifdef __cplusplus
}
#endif
#endif
