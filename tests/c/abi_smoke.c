/* Plain-C caller of the C ABI (what a cgo / Rust-FFI / JNI binding sees): creates a context, runs one tiny Beaver mask in
 * host-buffer mode and prints the result; without a GPU it must get ARKMPC_ERR_NO_DEVICE and exit 3 -- no CPU fallback. */
#include <stdio.h>
#include <string.h>
#include "arkmpc.h"

int main(void) {
    arkmpc_ctx* ctx = NULL;
    int rc = arkmpc_ctx_create(ARKMPC_BN254_FR, 0, &ctx);
    if (rc == ARKMPC_ERR_NO_DEVICE) { printf("no device: status %d, version %s\n", rc, arkmpc_version()); return 3; }
    if (rc != ARKMPC_OK) { printf("ctx_create failed: %d\n", rc); return 1; }
    if (arkmpc_ctx_set_host_buffers(ctx, 1) != ARKMPC_OK) return 1;
    /* x - a and y - b on the share halves, all operands small canonical-looking limbs (valid residues) */
    uint64_t x[8] = {9, 0, 0, 0, 1, 0, 0, 0}, y[8] = {7, 0, 0, 0, 2, 0, 0, 0}, a[8] = {4, 0, 0, 0, 3, 0, 0, 0}, b[8] = {5, 0, 0, 0, 4, 0, 0, 0};
    uint64_t de[8];
    memset(de, 0xff, sizeof de);
    rc = arkmpc_beaver_mask(ctx, 1, x, y, a, b, de);
    if (rc != ARKMPC_OK) { printf("beaver_mask failed: %d %s\n", rc, arkmpc_last_error(ctx)); return 1; }
    printf("d = %llu e = %llu\n", (unsigned long long)de[0], (unsigned long long)de[4]);
    arkmpc_ctx_destroy(ctx);
    return (de[0] == 5 && de[4] == 2 && de[1] == 0 && de[7] == 0) ? 0 : 1;
}
