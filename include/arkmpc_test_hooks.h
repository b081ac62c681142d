/*
 * arkmpc_test_hooks.h -- entry points the library exports for its OWN tests.  Not part of the drop-in boundary (include/arkmpc.h): nothing a
 * caller of the reference's API needs, no stability promise, absent from the generated Rust FFI.  Declared here so that every exported
 * `arkmpc_*` symbol is declared in exactly one of the two headers (tests/test_abi_cpu.py checks the export table against both).
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
/* process-wide count of streaming-session phases that ran as zero-copy kernels on the caller's pinned vectors (csrc/arkmpc_stream.inc):
 * out[0] = phase 1 (arkmpc_hostmul_begin), out[1] = phase 2 (arkmpc_hostmul_finish).  Lets the tests assert WHICH path produced a result. */
int arkmpc_test_hostmul_zero_copy_phases(uint64_t out[2]);
/* throws std::bad_alloc from inside an entry point's body: the process must end with the library's message on stderr (abort), never unwind
 * into the caller (csrc/arkmpc_internal.hpp CtxGuard).  Run it in a child process. */
int arkmpc_test_throw_inside(arkmpc_ctx* ctx);
#ifdef __cplusplus
}
#endif
#endif
