/*
 * ark_oracle.c -- CPU restatement (plain C, scalar, one thread) of the ark-mpc hot path.
 * TEST INFRASTRUCTURE ONLY -- see ark_oracle.h for the rules and the parity status
 * ("parity unpinned" by reference-held vectors; pinned by Python big-int + published constants).
 *
 * Every function cites the reference lines it restates (paths relative to /root/reference).
 * The field arithmetic itself lives in arkworks 0.4 (ark-ff `Fp256<MontBackend<_,4>>`), which is
 * NOT vendored in the reference; what is restated here is its published algorithm: 4x64-bit-limb
 * Montgomery (CIOS) multiplication with R = 2^256 and fully-reduced canonical outputs in [0, p).
 */
#include "ark_oracle.h"
#include <string.h>
#include <stdlib.h>
#include <pthread.h>

typedef unsigned __int128 u128;

/* ------------------------------------------------------------------------------------------
 * Field tables.  Only the moduli are written down; R, R^2 and INV are derived at first use and
 * are cross-checked against arkworks' published constants in tests/test_oracle_field.py.
 * ---------------------------------------------------------------------------------------- */
static const u64 MODULI[ORA_NFIELDS][4] = {
    /* BN254 Fr  0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001 */
    {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL},
    /* BLS12-381 Fr 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001 */
    {0xffffffff00000001ULL, 0x53bda402fffe5bfeULL, 0x3339d80809a1d805ULL, 0x73eda753299d7d48ULL},
    /* Curve25519 Fr (ed25519 group order l = 2^252 + 27742317777372353535851937790883648493) */
    {0x5812631a5cf5d3edULL, 0x14def9dea2f79cd6ULL, 0x0000000000000000ULL, 0x1000000000000000ULL},
    /* BN254 Fq  0x30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47 */
    {0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL},
    /* Curve25519 Fq  2^255 - 19 */
    {0xffffffffffffffedULL, 0xffffffffffffffffULL, 0xffffffffffffffffULL, 0x7fffffffffffffffULL},
};

static ora_field FIELDS[ORA_NFIELDS];
static int FIELDS_READY = 0;

static int geq(const u64 a[4], const u64 b[4]) {
    for (int i = 3; i >= 0; --i) {
        if (a[i] > b[i]) return 1;
        if (a[i] < b[i]) return 0;
    }
    return 1;
}
static u64 add4(const u64 a[4], const u64 b[4], u64 out[4]) {
    u128 c = 0;
    for (int i = 0; i < 4; ++i) { c += (u128)a[i] + b[i]; out[i] = (u64)c; c >>= 64; }
    return (u64)c;
}
static u64 sub4(const u64 a[4], const u64 b[4], u64 out[4]) {
    u64 borrow = 0;
    for (int i = 0; i < 4; ++i) {
        u128 d = (u128)a[i] - b[i] - borrow;
        out[i] = (u64)d; borrow = (u64)(d >> 64) & 1;
    }
    return borrow;
}
static int is_zero4(const u64 a[4]) { return (a[0] | a[1] | a[2] | a[3]) == 0; }

static void raw_add_mod(const u64 p[4], const u64 a[4], const u64 b[4], u64 out[4]) {
    u64 t[4]; u64 carry = add4(a, b, t);
    if (carry || geq(t, p)) sub4(t, p, t);
    memcpy(out, t, 32);
}

static void init_fields(void) {
    if (FIELDS_READY) return;
    for (int f = 0; f < ORA_NFIELDS; ++f) {
        ora_field* F = &FIELDS[f];
        memcpy(F->p, MODULI[f], 32);
        /* bit length */
        int bits = 256; while (bits > 0 && !((F->p[(bits - 1) / 64] >> ((bits - 1) % 64)) & 1)) --bits;
        F->bits = bits;
        /* R = 2^256 mod p and R2 = 2^512 mod p by repeated doubling of 1 */
        u64 v[4] = {1, 0, 0, 0};
        for (int i = 0; i < 256; ++i) raw_add_mod(F->p, v, v, v);
        memcpy(F->r, v, 32);
        for (int i = 0; i < 256; ++i) raw_add_mod(F->p, v, v, v);
        memcpy(F->r2, v, 32);
        /* inv = -p^{-1} mod 2^64 (Newton) */
        u64 x = 1; for (int i = 0; i < 7; ++i) x *= 2 - F->p[0] * x;
        F->inv = (u64)0 - x;
    }
    FIELDS_READY = 1;
}

const ora_field* ora_get_field(int field_id) {
    init_fields();
    if (field_id < 0 || field_id >= ORA_NFIELDS) return 0;
    return &FIELDS[field_id];
}

/* ---- ark-ff Fp ops: scalar.rs:210-267 delegate to these ---- */
void ora_fp_add(const ora_field* f, const u64 a[4], const u64 b[4], u64 out[4]) { raw_add_mod(f->p, a, b, out); }
void ora_fp_sub(const ora_field* f, const u64 a[4], const u64 b[4], u64 out[4]) {
    u64 t[4];
    if (sub4(a, b, t)) add4(t, f->p, t);
    memcpy(out, t, 32);
}
void ora_fp_neg(const ora_field* f, const u64 a[4], u64 out[4]) {
    if (is_zero4(a)) { memset(out, 0, 32); return; }
    u64 t[4]; sub4(f->p, a, t); memcpy(out, t, 32);
}
/* Montgomery multiplication, CIOS, as published for ark-ff MontBackend (out = a*b*R^{-1} mod p) */
void ora_fp_mul(const ora_field* f, const u64 a[4], const u64 b[4], u64 out[4]) {
    u64 t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; ++i) {
        u128 carry = 0, s;
        for (int j = 0; j < 4; ++j) { s = (u128)a[j] * b[i] + t[j] + carry; t[j] = (u64)s; carry = s >> 64; }
        s = (u128)t[4] + carry; t[4] = (u64)s; t[5] = (u64)(s >> 64);
        u64 m = t[0] * f->inv;
        s = (u128)m * f->p[0] + t[0]; carry = s >> 64;
        for (int j = 1; j < 4; ++j) { s = (u128)m * f->p[j] + t[j] + carry; t[j - 1] = (u64)s; carry = s >> 64; }
        s = (u128)t[4] + carry; t[3] = (u64)s; t[4] = t[5] + (u64)(s >> 64); t[5] = 0;
    }
    if (t[4] || geq(t, f->p)) sub4(t, f->p, t);
    memcpy(out, t, 32);
}
static void fp_sqr(const ora_field* f, const u64 a[4], u64 out[4]) { ora_fp_mul(f, a, a, out); }

void ora_fp_from_canonical(const ora_field* f, const u64 a[4], u64 out[4]) {
    u64 t[4]; memcpy(t, a, 32);
    while (geq(t, f->p)) sub4(t, f->p, t);
    ora_fp_mul(f, t, f->r2, out);
}
void ora_fp_to_canonical(const ora_field* f, const u64 a[4], u64 out[4]) {
    static const u64 one[4] = {1, 0, 0, 0};
    ora_fp_mul(f, a, one, out);
}
/* Fermat inverse a^(p-2); arkworks' `inverse()` returns the same unique residue (scalar.rs:83-85) */
void ora_fp_inv(const ora_field* f, const u64 a[4], u64 out[4]) {
    u64 e[4], two[4] = {2, 0, 0, 0}; sub4(f->p, two, e);
    u64 acc[4]; memcpy(acc, f->r, 32);
    for (int i = 255; i >= 0; --i) {
        fp_sqr(f, acc, acc);
        if ((e[i / 64] >> (i % 64)) & 1) ora_fp_mul(f, acc, a, acc);
    }
    memcpy(out, acc, 32);
}
/* scalar.rs:118-127: big-endian, left-padded to n_bytes_field = ceil(bits/8) = 32 for all four fields */
void ora_fp_to_bytes_be(const ora_field* f, const u64 a[4], unsigned char out[32]) {
    u64 c[4]; ora_fp_to_canonical(f, a, c);
    for (int i = 0; i < 4; ++i)
        for (int b = 0; b < 8; ++b) out[31 - (i * 8 + b)] = (unsigned char)(c[i] >> (8 * b));
}
/* scalar.rs:109-112: integer value of the big-endian bytes, reduced mod p (Horner, base 256) */
void ora_fp_from_be_bytes_mod_order(const ora_field* f, const unsigned char* bytes, size_t len, u64 out[4]) {
    u64 acc[4] = {0, 0, 0, 0}, m256[4], c256[4] = {256, 0, 0, 0};
    ora_fp_from_canonical(f, c256, m256);
    for (size_t i = 0; i < len; ++i) {
        u64 b[4] = {bytes[i], 0, 0, 0}, bm[4];
        ora_fp_from_canonical(f, b, bm);
        ora_fp_mul(f, acc, m256, acc);
        ora_fp_add(f, acc, bm, acc);
    }
    memcpy(out, acc, 32);
}

void ora_fp_batch_from_canonical(int fid, size_t n, const u64* in, u64* out) {
    const ora_field* f = ora_get_field(fid);
    for (size_t i = 0; i < n; ++i) ora_fp_from_canonical(f, in + 4 * i, out + 4 * i);
}
void ora_fp_batch_to_canonical(int fid, size_t n, const u64* in, u64* out) {
    const ora_field* f = ora_get_field(fid);
    for (size_t i = 0; i < n; ++i) ora_fp_to_canonical(f, in + 4 * i, out + 4 * i);
}
void ora_scalar_batch_add(int fid, size_t n, const u64* a, const u64* b, u64* out) {
    const ora_field* f = ora_get_field(fid);
    for (size_t i = 0; i < n; ++i) ora_fp_add(f, a + 4 * i, b + 4 * i, out + 4 * i);
}
void ora_scalar_batch_sub(int fid, size_t n, const u64* a, const u64* b, u64* out) {
    const ora_field* f = ora_get_field(fid);
    for (size_t i = 0; i < n; ++i) ora_fp_sub(f, a + 4 * i, b + 4 * i, out + 4 * i);
}
/* scalar_result.rs:257-278 */
void ora_scalar_batch_mul(int fid, size_t n, const u64* a, const u64* b, u64* out) {
    const ora_field* f = ora_get_field(fid);
    for (size_t i = 0; i < n; ++i) ora_fp_mul(f, a + 4 * i, b + 4 * i, out + 4 * i);
}
/* gadgets.rs:131-137: prefix = prefix * blinded_term, sequentially */
void ora_scalar_prefix_product(int fid, size_t n, const u64* a, u64* out) {
    const ora_field* f = ora_get_field(fid);
    u64 run[4]; memcpy(run, f->r, 32);
    for (size_t i = 0; i < n; ++i) { ora_fp_mul(f, run, a + 4 * i, run); memcpy(out + 4 * i, run, 32); }
}
/* Iterator::sum / Product for ScalarResult (scalar_result.rs:325-338 `args.map(Scalar::from).product()`; scalar.rs Sum / Product impls):
 * a left-to-right fold from 0 / 1 */
void ora_scalar_sum(int fid, size_t n, const u64* a, u64 out[4]) {
    const ora_field* f = ora_get_field(fid);
    u64 run[4] = {0, 0, 0, 0};
    for (size_t i = 0; i < n; ++i) ora_fp_add(f, run, a + 4 * i, run);
    memcpy(out, run, 32);
}
void ora_scalar_product(int fid, size_t n, const u64* a, u64 out[4]) {
    const ora_field* f = ora_get_field(fid);
    u64 run[4]; memcpy(run, f->r, 32);
    for (size_t i = 0; i < n; ++i) ora_fp_mul(f, run, a + 4 * i, run);
    memcpy(out, run, 32);
}
/* Sum for ScalarShare (share.rs:103-111): unzip into shares and macs, sum each; the gate of Sum for AuthenticatedScalarResult
 * (authenticated_scalar.rs:563-575) */
void ora_share_sum(int fid, size_t n, const u64* shares, u64 out[8]) {
    const ora_field* f = ora_get_field(fid);
    u64 s[4] = {0, 0, 0, 0}, m[4] = {0, 0, 0, 0};
    for (size_t i = 0; i < n; ++i) { ora_fp_add(f, s, shares + 8 * i, s); ora_fp_add(f, m, shares + 8 * i + 4, m); }
    memcpy(out, s, 32); memcpy(out + 4, m, 32);
}
/* scalar.rs:93-100 -> ark_ff::batch_inversion: non-zero elements inverted, zeros unchanged */
void ora_scalar_batch_inverse(int fid, size_t n, const u64* a, u64* out) {
    const ora_field* f = ora_get_field(fid);
    for (size_t i = 0; i < n; ++i) {
        if (is_zero4(a + 4 * i)) memset(out + 4 * i, 0, 32);
        else ora_fp_inv(f, a + 4 * i, out + 4 * i);
    }
}
void ora_scalar_batch_neg(int fid, size_t n, const u64* a, u64* out) {
    const ora_field* f = ora_get_field(fid);
    for (size_t i = 0; i < n; ++i) ora_fp_neg(f, a + 4 * i, out + 4 * i);
}

/* ------------------------------------------------------------------------------------------
 * ScalarShare (share.rs:72-133).  A share is 8 limbs: share[0..4), mac[4..8).
 * ---------------------------------------------------------------------------------------- */
/* share.rs:85-91 */
static void share_add(const ora_field* f, const u64* a, const u64* b, u64* out) {
    ora_fp_add(f, a, b, out); ora_fp_add(f, a + 4, b + 4, out + 4);
}
/* share.rs:115-121 */
static void share_neg(const ora_field* f, const u64* a, u64* out) { ora_fp_neg(f, a, out); ora_fp_neg(f, a + 4, out + 4); }
/* share.rs:95-101: self + (-rhs) */
static void share_sub(const ora_field* f, const u64* a, const u64* b, u64* out) {
    u64 nb[8]; share_neg(f, b, nb); share_add(f, a, nb, out);
}
/* share.rs:125-131 */
static void share_mul_public(const ora_field* f, const u64* a, const u64* s, u64* out) {
    ora_fp_mul(f, a, s, out); ora_fp_mul(f, a + 4, s, out + 4);
}
/* share.rs:74-77: share += rhs iff PARTY0; mac += mac_key * rhs */
static void share_add_public(const ora_field* f, int party, const u64* key, const u64* a, const u64* rhs, u64* out) {
    u64 km[4], sh[4];
    if (party == 0) ora_fp_add(f, a, rhs, sh); else memcpy(sh, a, 32);
    ora_fp_mul(f, key, rhs, km);
    ora_fp_add(f, a + 4, km, out + 4);
    memcpy(out, sh, 32);
}
/* share.rs:80-82 */
static void share_sub_public(const ora_field* f, int party, const u64* key, const u64* a, const u64* rhs, u64* out) {
    u64 nr[4]; ora_fp_neg(f, rhs, nr); share_add_public(f, party, key, a, nr, out);
}

void ora_share_batch_add(int fid, size_t n, const u64* a, const u64* b, u64* out) {
    const ora_field* f = ora_get_field(fid);
    for (size_t i = 0; i < n; ++i) share_add(f, a + 8 * i, b + 8 * i, out + 8 * i);
}
void ora_share_batch_sub(int fid, size_t n, const u64* a, const u64* b, u64* out) {
    const ora_field* f = ora_get_field(fid);
    for (size_t i = 0; i < n; ++i) share_sub(f, a + 8 * i, b + 8 * i, out + 8 * i);
}
void ora_share_batch_neg(int fid, size_t n, const u64* a, u64* out) {
    const ora_field* f = ora_get_field(fid);
    for (size_t i = 0; i < n; ++i) share_neg(f, a + 8 * i, out + 8 * i);
}
void ora_share_batch_add_public(int fid, size_t n, int party, const u64 key[4], const u64* a, const u64* pub, u64* out) {
    const ora_field* f = ora_get_field(fid);
    for (size_t i = 0; i < n; ++i) share_add_public(f, party, key, a + 8 * i, pub + 4 * i, out + 8 * i);
}
void ora_share_batch_sub_public(int fid, size_t n, int party, const u64 key[4], const u64* a, const u64* pub, u64* out) {
    const ora_field* f = ora_get_field(fid);
    for (size_t i = 0; i < n; ++i) share_sub_public(f, party, key, a + 8 * i, pub + 4 * i, out + 8 * i);
}
void ora_share_batch_mul_public(int fid, size_t n, const u64* a, const u64* pub, u64* out) {
    const ora_field* f = ora_get_field(fid);
    for (size_t i = 0; i < n; ++i) share_mul_public(f, a + 8 * i, pub + 4 * i, out + 8 * i);
}

/* ------------------------------------------------------------------------------------------
 * Hot path, flat-batched.
 * ---------------------------------------------------------------------------------------- */
/* authenticated_scalar.rs:863-868 -- masked_lhs = a - beaver_a, masked_rhs = b - beaver_b,
 * all_masks = lhs || rhs; open_batch sends only `.share()` (:141-144), so only shares are produced */
void ora_beaver_mask(int fid, size_t n, const u64* x, const u64* y, const u64* a, const u64* b, u64* out_de) {
    const ora_field* f = ora_get_field(fid);
    for (size_t i = 0; i < n; ++i) {
        ora_fp_sub(f, x + 8 * i, a + 8 * i, out_de + 4 * i);
        ora_fp_sub(f, y + 8 * i, b + 8 * i, out_de + 4 * (n + i));
    }
}
/* authenticated_scalar.rs:161-171 */
void ora_open_combine(int fid, size_t n, const u64* mine, const u64* peer, u64* out) {
    ora_scalar_batch_add(fid, n, mine, peer, out);
}
/* authenticated_scalar.rs:871-878 (batch) == :835-840 (single): de + d[b] + e[a] + [c] */
void ora_beaver_finish(int fid, size_t n, int party, const u64 key[4], const u64* d, const u64* e, const u64* a,
                       const u64* b, const u64* c, u64* out) {
    const ora_field* f = ora_get_field(fid);
    for (size_t i = 0; i < n; ++i) {
        u64 de[4], db[8], ea[8], t[8], u[8];
        ora_fp_mul(f, d + 4 * i, e + 4 * i, de);                /* ScalarResult::batch_mul(d, e)        :871 */
        share_mul_public(f, b + 8 * i, d + 4 * i, db);          /* batch_mul_public(beaver_b, d)        :872 */
        share_mul_public(f, a + 8 * i, e + 4 * i, ea);          /* batch_mul_public(beaver_a, e)        :873 */
        share_add_public(f, party, key, db, de, t);             /* batch_add_public(db, de)             :876 */
        share_add(f, ea, c + 8 * i, u);                         /* batch_add(ea, beaver_c)              :877 */
        share_add(f, t, u, out + 8 * i);                        /* batch_add(de_plus_db, ea_plus_c)     :878 */
    }
}
/* The literal pass structure of authenticated_scalar.rs:848-879 for ONE party, as separate sweeps
 * over memory (what the reference's 9 GateBatch ops do), with the peer's d||e shares supplied.
 * scratch: 8*n shares (64*n u64). */
void ora_batch_mul_9pass_local(int fid, size_t n, int party, const u64 key[4], const u64* x, const u64* y, const u64* a,
                               const u64* b, const u64* c, const u64* peer_de, u64* my_de, u64* out, u64* scratch) {
    u64* masked_lhs = scratch;            /* n shares */
    u64* masked_rhs = scratch + 8 * n;    /* n shares */
    u64* opened = scratch + 16 * n;       /* 2n scalars = n shares worth */
    u64* de = scratch + 24 * n;           /* n scalars */
    u64* db = scratch + 32 * n;
    u64* ea = scratch + 40 * n;
    u64* t = scratch + 48 * n;
    u64* u = scratch + 56 * n;
    ora_share_batch_sub(fid, n, x, a, masked_lhs);                              /* pass 1 :863 */
    ora_share_batch_sub(fid, n, y, b, masked_rhs);                              /* pass 2 :864 */
    for (size_t i = 0; i < n; ++i) {                                            /* network op closure :141-145 */
        memcpy(my_de + 4 * i, masked_lhs + 8 * i, 32);
        memcpy(my_de + 4 * (n + i), masked_rhs + 8 * i, 32);
    }
    ora_scalar_batch_add(fid, 2 * n, my_de, peer_de, opened);                   /* pass 3 :161-171 */
    ora_scalar_batch_mul(fid, n, opened, opened + 4 * n, de);                   /* pass 4 :871 */
    ora_share_batch_mul_public(fid, n, b, opened, db);                          /* pass 5 :872 */
    ora_share_batch_mul_public(fid, n, a, opened + 4 * n, ea);                  /* pass 6 :873 */
    ora_share_batch_add_public(fid, n, party, key, db, de, t);                  /* pass 7 :876 */
    ora_share_batch_add(fid, n, ea, c, u);                                      /* pass 8 :877 */
    ora_share_batch_add(fid, n, t, u, out);                                     /* pass 9 :878 */
}
/* authenticated_scalar.rs:299-311: mac_key * value - share.mac() */
void ora_mac_check_shares(int fid, size_t n, const u64 key[4], const u64* opened, const u64* shares, u64* out) {
    const ora_field* f = ora_get_field(fid);
    for (size_t i = 0; i < n; ++i) {
        u64 kv[4]; ora_fp_mul(f, key, opened + 4 * i, kv);
        ora_fp_sub(f, kv, shares + 8 * i + 4, out + 4 * i);
    }
}
/* authenticated_scalar.rs:218-219 */
int ora_mac_verify(int fid, size_t n, const u64* mine, const u64* peer) {
    const ora_field* f = ora_get_field(fid);
    for (size_t i = 0; i < n; ++i) {
        u64 s[4]; ora_fp_add(f, mine + 4 * i, peer + 4 * i, s);
        if (!is_zero4(s)) return 0;
    }
    return 1;
}

/* ------------------------------------------------------------------------------------------
 * SHA3-256 (FIPS 202) -- the `sha3 = "0.10"` crate's Sha3_256 (commitment.rs:36-40, :79-83).
 * ---------------------------------------------------------------------------------------- */
static const u64 KECCAK_RC[24] = {
    0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL, 0x000000000000808bULL,
    0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL, 0x000000000000008aULL, 0x0000000000000088ULL,
    0x0000000080008009ULL, 0x000000008000000aULL, 0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL,
    0x8000000000008003ULL, 0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
    0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
static const int KECCAK_ROT[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};
static u64 rotl64(u64 x, int s) { return s ? (x << s) | (x >> (64 - s)) : x; }
static void keccak_f(u64 st[25]) {
    for (int round = 0; round < 24; ++round) {
        u64 C[5], D[5], B[25];
        for (int x = 0; x < 5; ++x) C[x] = st[x] ^ st[x + 5] ^ st[x + 10] ^ st[x + 15] ^ st[x + 20];
        for (int x = 0; x < 5; ++x) D[x] = C[(x + 4) % 5] ^ rotl64(C[(x + 1) % 5], 1);
        for (int i = 0; i < 25; ++i) st[i] ^= D[i % 5];
        for (int x = 0; x < 5; ++x)
            for (int y = 0; y < 5; ++y) B[y + 5 * ((2 * x + 3 * y) % 5)] = rotl64(st[x + 5 * y], KECCAK_ROT[x + 5 * y]);
        for (int y = 0; y < 5; ++y)
            for (int x = 0; x < 5; ++x) st[x + 5 * y] = B[x + 5 * y] ^ ((~B[(x + 1) % 5 + 5 * y]) & B[(x + 2) % 5 + 5 * y]);
        st[0] ^= KECCAK_RC[round];
    }
}
typedef struct { u64 st[25]; unsigned char buf[136]; size_t fill; } sha3_ctx;
static void sha3_init(sha3_ctx* c) { memset(c, 0, sizeof(*c)); }
static void sha3_absorb_block(sha3_ctx* c, const unsigned char* blk) {
    for (int i = 0; i < 17; ++i) {
        u64 w = 0; for (int b = 0; b < 8; ++b) w |= (u64)blk[8 * i + b] << (8 * b);
        c->st[i] ^= w;
    }
    keccak_f(c->st);
}
static void sha3_update(sha3_ctx* c, const unsigned char* m, size_t len) {
    while (len) {
        size_t take = 136 - c->fill; if (take > len) take = len;
        memcpy(c->buf + c->fill, m, take); c->fill += take; m += take; len -= take;
        if (c->fill == 136) { sha3_absorb_block(c, c->buf); c->fill = 0; }
    }
}
static void sha3_final(sha3_ctx* c, unsigned char out[32]) {
    memset(c->buf + c->fill, 0, 136 - c->fill);
    c->buf[c->fill] ^= 0x06; c->buf[135] ^= 0x80;
    sha3_absorb_block(c, c->buf);
    for (int i = 0; i < 4; ++i) for (int b = 0; b < 8; ++b) out[8 * i + b] = (unsigned char)(c->st[i] >> (8 * b));
}
void ora_sha3_256(const unsigned char* msg, size_t len, unsigned char out[32]) {
    sha3_ctx c; sha3_init(&c); sha3_update(&c, msg, len); sha3_final(&c, out);
}
/* commitment.rs:71-86 (and verify :30-43): H(BE(v_0) || ... || BE(v_{n-1}) || BE(blinder)) */
void ora_commit_scalars(int fid, size_t n, const u64* values, const u64 blinder[4], u64 out[4]) {
    const ora_field* f = ora_get_field(fid);
    sha3_ctx c; sha3_init(&c);
    unsigned char be[32], dig[32];
    for (size_t i = 0; i < n; ++i) { ora_fp_to_bytes_be(f, values + 4 * i, be); sha3_update(&c, be, 32); }
    ora_fp_to_bytes_be(f, blinder, be); sha3_update(&c, be, 32);
    sha3_final(&c, dig);
    ora_fp_from_be_bytes_mod_order(f, dig, 32, out);
}
void ora_commit_bytes(int fid, const unsigned char* bytes, size_t len, const u64 blinder[4], u64 out[4]) {
    const ora_field* f = ora_get_field(fid);
    sha3_ctx c; sha3_init(&c);
    unsigned char be[32], dig[32];
    sha3_update(&c, bytes, len);
    ora_fp_to_bytes_be(f, blinder, be); sha3_update(&c, be, 32);
    sha3_final(&c, dig);
    ora_fp_from_be_bytes_mod_order(f, dig, 32, out);
}

/* ------------------------------------------------------------------------------------------
 * BN254 G1: y^2 = x^3 + 3 over Fq, generator (1, 2), cofactor 1.  ark-ec short_weierstrass
 * `Projective{x,y,z}` holds Jacobian coordinates; identity = (1, 1, 0).  curve.rs:194-409 delegates
 * `+`, `-`, neg and scalar-mul to ark-ec; the published formulas are add-2007-bl / dbl-2009-l (a = 0).
 * ---------------------------------------------------------------------------------------- */
#define FQ (ora_get_field(ORA_BN254_FQ))
#define FR (ora_get_field(ORA_BN254_FR))

void ora_g1_identity(u64 out[12]) {
    const ora_field* q = FQ; memcpy(out, q->r, 32); memcpy(out + 4, q->r, 32); memset(out + 8, 0, 32);
}
void ora_g1_generator(u64 out[12]) {
    const ora_field* q = FQ; memcpy(out, q->r, 32);
    ora_fp_add(q, q->r, q->r, out + 4); memcpy(out + 8, q->r, 32);
}
int ora_g1_is_identity(const u64 a[12]) { return is_zero4(a + 8); }
void ora_g1_neg(const u64 a[12], u64 out[12]) {
    const ora_field* q = FQ; u64 ny[4]; ora_fp_neg(q, a + 4, ny);
    memcpy(out, a, 32); memcpy(out + 4, ny, 32); memmove(out + 8, a + 8, 32);
}
void ora_g1_double(const u64 p[12], u64 out[12]) {
    const ora_field* q = FQ;
    if (ora_g1_is_identity(p)) { memmove(out, p, 96); return; }
    const u64 *X = p, *Y = p + 4, *Z = p + 8;
    u64 A[4], B[4], C[4], D[4], E[4], F[4], t[4], X3[4], Y3[4], Z3[4];
    fp_sqr(q, X, A); fp_sqr(q, Y, B); fp_sqr(q, B, C);
    ora_fp_add(q, X, B, t); fp_sqr(q, t, t); ora_fp_sub(q, t, A, t); ora_fp_sub(q, t, C, t); ora_fp_add(q, t, t, D);
    ora_fp_add(q, A, A, E); ora_fp_add(q, E, A, E);
    fp_sqr(q, E, F);
    ora_fp_sub(q, F, D, X3); ora_fp_sub(q, X3, D, X3);
    ora_fp_mul(q, Y, Z, Z3); ora_fp_add(q, Z3, Z3, Z3);
    ora_fp_sub(q, D, X3, t); ora_fp_mul(q, E, t, Y3);
    ora_fp_add(q, C, C, t); ora_fp_add(q, t, t, t); ora_fp_add(q, t, t, t);
    ora_fp_sub(q, Y3, t, Y3);
    memcpy(out, X3, 32); memcpy(out + 4, Y3, 32); memcpy(out + 8, Z3, 32);
}
void ora_g1_add(const u64 p1[12], const u64 p2[12], u64 out[12]) {
    const ora_field* q = FQ;
    if (ora_g1_is_identity(p1)) { memmove(out, p2, 96); return; }
    if (ora_g1_is_identity(p2)) { memmove(out, p1, 96); return; }
    const u64 *X1 = p1, *Y1 = p1 + 4, *Z1 = p1 + 8, *X2 = p2, *Y2 = p2 + 4, *Z2 = p2 + 8;
    u64 Z1Z1[4], Z2Z2[4], U1[4], U2[4], S1[4], S2[4], H[4], I[4], J[4], r[4], V[4], t[4], X3[4], Y3[4], Z3[4];
    fp_sqr(q, Z1, Z1Z1); fp_sqr(q, Z2, Z2Z2);
    ora_fp_mul(q, X1, Z2Z2, U1); ora_fp_mul(q, X2, Z1Z1, U2);
    ora_fp_mul(q, Y1, Z2, S1); ora_fp_mul(q, S1, Z2Z2, S1);
    ora_fp_mul(q, Y2, Z1, S2); ora_fp_mul(q, S2, Z1Z1, S2);
    if (memcmp(U1, U2, 32) == 0) {
        if (memcmp(S1, S2, 32) == 0) { ora_g1_double(p1, out); return; }
        ora_g1_identity(out); return;
    }
    ora_fp_sub(q, U2, U1, H);
    ora_fp_add(q, H, H, I); fp_sqr(q, I, I);
    ora_fp_mul(q, H, I, J);
    ora_fp_sub(q, S2, S1, r); ora_fp_add(q, r, r, r);
    ora_fp_mul(q, U1, I, V);
    fp_sqr(q, r, X3); ora_fp_sub(q, X3, J, X3); ora_fp_sub(q, X3, V, X3); ora_fp_sub(q, X3, V, X3);
    ora_fp_sub(q, V, X3, t); ora_fp_mul(q, r, t, Y3);
    ora_fp_mul(q, S1, J, t); ora_fp_add(q, t, t, t); ora_fp_sub(q, Y3, t, Y3);
    ora_fp_add(q, Z1, Z2, Z3); fp_sqr(q, Z3, Z3); ora_fp_sub(q, Z3, Z1Z1, Z3); ora_fp_sub(q, Z3, Z2Z2, Z3);
    ora_fp_mul(q, Z3, H, Z3);
    memcpy(out, X3, 32); memcpy(out + 4, Y3, 32); memcpy(out + 8, Z3, 32);
}
/* curve.rs:403-409 `self.0 * rhs.0`: the group element [s]P; MSB-first double-and-add */
void ora_g1_scalar_mul(const u64 pt[12], const u64 scalar_mont[4], u64 out[12]) {
    u64 s[4]; ora_fp_to_canonical(FR, scalar_mont, s);
    u64 acc[12]; ora_g1_identity(acc);
    for (int i = 255; i >= 0; --i) {
        ora_g1_double(acc, acc);
        if ((s[i / 64] >> (i % 64)) & 1) ora_g1_add(acc, pt, acc);
    }
    memcpy(out, acc, 96);
}
int ora_g1_to_affine(const u64 a[12], u64 out_xy[8]) {
    const ora_field* q = FQ;
    if (ora_g1_is_identity(a)) { memset(out_xy, 0, 64); return 1; }
    u64 zi[4], zi2[4], zi3[4];
    ora_fp_inv(q, a + 8, zi); fp_sqr(q, zi, zi2); ora_fp_mul(q, zi2, zi, zi3);
    ora_fp_mul(q, a, zi2, out_xy); ora_fp_mul(q, a + 4, zi3, out_xy + 4);
    return 0;
}
int ora_g1_eq(const u64 a[12], const u64 b[12]) {
    u64 xa[8], xb[8]; int ia = ora_g1_to_affine(a, xa), ib = ora_g1_to_affine(b, xb);
    if (ia || ib) return ia == ib;
    return memcmp(xa, xb, 64) == 0;
}
/* ark-serialize compressed SW point: x little-endian, flags in the top bits of the last byte:
 * bit 7 = y is the lexicographically larger root (y > -y), bit 6 = point at infinity (x = 0) */
void ora_g1_to_bytes(const u64 a[12], unsigned char out[32]) {
    const ora_field* q = FQ; u64 xy[8];
    memset(out, 0, 32);
    if (ora_g1_to_affine(a, xy)) { out[31] |= 0x40; return; }
    u64 xc[4], yc[4], nyc[4], ny[4];
    ora_fp_to_canonical(q, xy, xc); ora_fp_to_canonical(q, xy + 4, yc);
    ora_fp_neg(q, xy + 4, ny); ora_fp_to_canonical(q, ny, nyc);
    for (int i = 0; i < 4; ++i) for (int b = 0; b < 8; ++b) out[8 * i + b] = (unsigned char)(xc[i] >> (8 * b));
    int y_gt_neg = geq(yc, nyc) && memcmp(yc, nyc, 32) != 0;
    if (y_gt_neg) out[31] |= 0x80;
}
/* CurvePoint::from_bytes (curve.rs:110-114) = deserialize_compressed with validation: x must be canonical (< q), at most one
 * flag bit set; infinity flag -> identity; otherwise y = sqrt(x^3 + 3) must exist (q = 3 mod 4: y = rhs^((q+1)/4)) and the
 * root selected is the larger one iff bit 7 is set (the inverse of ora_g1_to_bytes).  Returns 1 if valid, 0 (and the
 * identity) if the bytes are not a point encoding.  BN254 G1 has cofactor 1, so on-curve implies in-subgroup. */
int ora_g1_from_bytes(const unsigned char in[32], u64 out[12]) {
    const ora_field* q = FQ;
    ora_g1_identity(out);
    const int neg = (in[31] >> 7) & 1, inf = (in[31] >> 6) & 1;
    if (neg && inf) return 0;
    u64 xc[4] = {0, 0, 0, 0};
    for (int i = 0; i < 32; ++i) { unsigned char b = in[i]; if (i == 31) b &= 0x3f; xc[i / 8] |= (u64)b << (8 * (i % 8)); }
    if (geq(xc, q->p)) return 0;
    if (inf) return 1;
    u64 x[4], rhs[4], three[4] = {3, 0, 0, 0}, t[4];
    ora_fp_from_canonical(q, xc, x);
    fp_sqr(q, x, t); ora_fp_mul(q, t, x, rhs); ora_fp_from_canonical(q, three, t); ora_fp_add(q, rhs, t, rhs);
    u64 e[4];
    {   /* q + 1 does not overflow 256 bits (q < 2^254) */
        unsigned __int128 cy = 1;
        for (int i = 0; i < 4; ++i) { cy += q->p[i]; e[i] = (u64)cy; cy >>= 64; }
        for (int i = 0; i < 4; ++i) e[i] = (e[i] >> 2) | (i < 3 ? e[i + 1] << 62 : 0);
    }
    u64 y[4]; memcpy(y, q->r, 32);
    for (int i = 255; i >= 0; --i) { fp_sqr(q, y, y); if ((e[i / 64] >> (i % 64)) & 1) ora_fp_mul(q, y, rhs, y); }
    fp_sqr(q, y, t);
    if (memcmp(t, rhs, 32) != 0) return 0;                  /* x^3 + 3 is not a square */
    u64 ny[4], yc[4], nyc[4];
    ora_fp_neg(q, y, ny); ora_fp_to_canonical(q, y, yc); ora_fp_to_canonical(q, ny, nyc);
    const int y_is_larger = geq(yc, nyc) && memcmp(yc, nyc, 32) != 0;
    memcpy(out, x, 32); memcpy(out + 4, (y_is_larger == neg) ? y : ny, 32); memcpy(out + 8, q->r, 32);
    return 1;
}
void ora_g1_batch_add(size_t n, const u64* a, const u64* b, u64* out) {
    for (size_t i = 0; i < n; ++i) ora_g1_add(a + 12 * i, b + 12 * i, out + 12 * i);
}
void ora_g1_batch_scalar_mul(size_t n, const u64* pts, const u64* scalars, u64* out) {
    for (size_t i = 0; i < n; ++i) ora_g1_scalar_mul(pts + 12 * i, scalars + 4 * i, out + 12 * i);
}
void ora_g1_batch_to_affine(size_t n, const u64* pts, u64* out_xy, unsigned char* is_inf) {
    for (size_t i = 0; i < n; ++i) is_inf[i] = (unsigned char)ora_g1_to_affine(pts + 12 * i, out_xy + 8 * i);
}
/* authenticated_curve.rs:796-805 / curve/share.rs:85-92: left fold of the group law */
void ora_g1_sum(size_t n, const u64* pts, size_t stride, u64 out[12]) {
    u64 acc[12]; ora_g1_identity(acc);
    for (size_t i = 0; i < n; ++i) ora_g1_add(acc, pts + stride * i, acc);
    memcpy(out, acc, 96);
}
/* CurvePoint::msm (curve.rs:549-560): sum_i scalars[i] * points[i].  The reference hands affine points and big-integer
 * scalars to ark-ec's VariableBaseMSM (a Pippenger bucket method); the group element is defined by the plain sum, which is
 * what this restates (scalar stride in u64 lets it read one column of a ScalarShare array). */
void ora_g1_msm(size_t n, const u64* pts, const u64* scalars, size_t scalar_stride, u64 out[12]) {
    u64 acc[12], t[12]; ora_g1_identity(acc);
    for (size_t i = 0; i < n; ++i) { ora_g1_scalar_mul(pts + 12 * i, scalars + scalar_stride * i, t); ora_g1_add(acc, t, acc); }
    memcpy(out, acc, 96);
}
/* CurvePoint::msm_authenticated (curve.rs:618-642): PointShare(msm(shares, points), msm(macs, points)) */
void ora_g1_msm_authenticated(size_t n, const u64* pts, const u64* scalar_shares, u64 out[24]) {
    ora_g1_msm(n, pts, scalar_shares, 8, out);
    ora_g1_msm(n, pts, scalar_shares + 4, 8, out + 12);
}
/* curve/share.rs:68-105 */
void ora_pointshare_batch_add(size_t n, const u64* a, const u64* b, u64* out) {
    for (size_t i = 0; i < n; ++i) {
        ora_g1_add(a + 24 * i, b + 24 * i, out + 24 * i);
        ora_g1_add(a + 24 * i + 12, b + 24 * i + 12, out + 24 * i + 12);
    }
}
void ora_pointshare_batch_neg(size_t n, const u64* a, u64* out) {
    for (size_t i = 0; i < 2 * n; ++i) ora_g1_neg(a + 12 * i, out + 12 * i);
}
void ora_pointshare_batch_sub(size_t n, const u64* a, const u64* b, u64* out) {
    for (size_t i = 0; i < 2 * n; ++i) { u64 nb[12]; ora_g1_neg(b + 12 * i, nb); ora_g1_add(a + 12 * i, nb, out + 12 * i); }
}
/* curve/share.rs:108-114 via authenticated_curve.rs:718-751 */
void ora_pointshare_batch_mul_public(size_t n, const u64* shares, const u64* scalars, u64* out) {
    for (size_t i = 0; i < n; ++i) {
        ora_g1_scalar_mul(shares + 24 * i, scalars + 4 * i, out + 24 * i);
        ora_g1_scalar_mul(shares + 24 * i + 12, scalars + 4 * i, out + 24 * i + 12);
    }
}
/* curve/share.rs:57-60 via authenticated_curve.rs:429-463 */
void ora_pointshare_batch_add_public(size_t n, int party, const u64 key[4], const u64* shares, const u64* pub, u64* out) {
    for (size_t i = 0; i < n; ++i) {
        u64 kp[12];
        ora_g1_scalar_mul(pub + 12 * i, key, kp);
        if (party == 0) ora_g1_add(shares + 24 * i, pub + 12 * i, out + 24 * i);
        else memmove(out + 24 * i, shares + 24 * i, 96);
        ora_g1_add(shares + 24 * i + 12, kp, out + 24 * i + 12);
    }
}
/* share.rs:135-141 with the generator, via authenticated_curve.rs:754-780 */
void ora_scalarshare_batch_mul_generator(size_t n, const u64* ss, u64* out) {
    u64 g[12]; ora_g1_generator(g);
    for (size_t i = 0; i < n; ++i) {
        ora_g1_scalar_mul(g, ss + 8 * i, out + 24 * i);
        ora_g1_scalar_mul(g, ss + 8 * i + 4, out + 24 * i + 12);
    }
}
/* curve.rs:483-517 batch_mul_authenticated: point * ScalarShare */
void ora_scalarshare_batch_mul_point(size_t n, const u64* ss, const u64* points, u64* out) {
    for (size_t i = 0; i < n; ++i) {
        ora_g1_scalar_mul(points + 12 * i, ss + 8 * i, out + 24 * i);
        ora_g1_scalar_mul(points + 12 * i, ss + 8 * i + 4, out + 24 * i + 12);
    }
}

/* ------------------------------------------------------------------------------------------
 * PartyIDBeaverSource (offline_prep.rs:88-170): a = 2, b = 3, c = 6; MAC key share = party id.
 * ---------------------------------------------------------------------------------------- */
static void small(const ora_field* f, u64 v, u64 out[4]) { u64 c[4] = {v, 0, 0, 0}; ora_fp_from_canonical(f, c, out); }
void ora_dummy_mac_key_share(int fid, int party, u64 out[4]) { small(ora_get_field(fid), (u64)party, out); } /* :108-110 */
/* offline_prep.rs:137-158 */
void ora_dummy_triples(int fid, int party, size_t n, u64* a, u64* b, u64* c) {
    const ora_field* f = ora_get_field(fid);
    u64 key[4], va[4], vb[4], vc[4], sa[4], sb[4], sc[4], ma[4], mb[4], mc[4];
    small(f, (u64)party, key); small(f, 2, va); small(f, 3, vb); small(f, 6, vc);
    ora_fp_mul(f, key, va, ma); ora_fp_mul(f, key, vb, mb); ora_fp_mul(f, key, vc, mc);
    if (party == 0) { small(f, 1, sa); small(f, 3, sb); small(f, 2, sc); }
    else { small(f, 1, sa); small(f, 0, sb); small(f, 4, sc); }
    for (size_t i = 0; i < n; ++i) {
        memcpy(a + 8 * i, sa, 32); memcpy(a + 8 * i + 4, ma, 32);
        memcpy(b + 8 * i, sb, 32); memcpy(b + 8 * i + 4, mb, 32);
        memcpy(c + 8 * i, sc, 32); memcpy(c + 8 * i + 4, mc, 32);
    }
}
/* offline_prep.rs:112-119: value 3, share = mac = party * 3 */
void ora_dummy_local_input_masks(int fid, int party, size_t n, u64* masks, u64* mask_shares) {
    const ora_field* f = ora_get_field(fid);
    u64 v[4], pv[4], pid[4]; small(f, 3, v); small(f, (u64)party, pid); ora_fp_mul(f, pid, v, pv);
    for (size_t i = 0; i < n; ++i) {
        memcpy(masks + 4 * i, v, 32); memcpy(mask_shares + 8 * i, pv, 32); memcpy(mask_shares + 8 * i + 4, pv, 32);
    }
}
/* offline_prep.rs:121-127: value = 3 * party, mac = party * value */
void ora_dummy_counterparty_input_masks(int fid, int party, size_t n, u64* mask_shares) {
    const ora_field* f = ora_get_field(fid);
    u64 v[4], three[4], pid[4], m[4]; small(f, 3, three); small(f, (u64)party, pid);
    ora_fp_mul(f, three, pid, v); ora_fp_mul(f, pid, v, m);
    for (size_t i = 0; i < n; ++i) { memcpy(mask_shares + 8 * i, v, 32); memcpy(mask_shares + 8 * i + 4, m, 32); }
}

/* ------------------------------------------------------------------------------------------
 * Multi-core form of the 9-pass batch_mul: static contiguous range split over `nthreads` pthreads, each
 * running ora_batch_mul_9pass_local on its range -- the upper bound for the reference's rayon executor
 * (fabric/executor/multi_threaded/executor.rs:208-217).  Used only by bench.py's cpu_baseline leg.
 * peer_de / my_de are d||e buffers of the FULL batch (d at [0,n), e at [n,2n)).
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    int fid, party; size_t n, lo, hi; const u64 *key, *x, *y, *a, *b, *c, *peer_de; u64 *my_de, *out; int rc;
} mt_job;
static void* mt_worker(void* arg) {
    mt_job* j = (mt_job*)arg;
    const size_t cnt = j->hi - j->lo, n = j->n, lo = j->lo;
    if (!cnt) return 0;
    u64* scratch = (u64*)malloc(64 * cnt * sizeof(u64));
    u64* pde = (u64*)malloc(8 * cnt * sizeof(u64));
    u64* mde = (u64*)malloc(8 * cnt * sizeof(u64));
    if (!scratch || !pde || !mde) { j->rc = 1; free(scratch); free(pde); free(mde); return 0; }
    memcpy(pde, j->peer_de + 4 * lo, 32 * cnt);
    memcpy(pde + 4 * cnt, j->peer_de + 4 * (n + lo), 32 * cnt);
    ora_batch_mul_9pass_local(j->fid, cnt, j->party, j->key, j->x + 8 * lo, j->y + 8 * lo, j->a + 8 * lo, j->b + 8 * lo,
                              j->c + 8 * lo, pde, mde, j->out + 8 * lo, scratch);
    memcpy(j->my_de + 4 * lo, mde, 32 * cnt);
    memcpy(j->my_de + 4 * (n + lo), mde + 4 * cnt, 32 * cnt);
    free(scratch); free(pde); free(mde);
    return 0;
}
int ora_batch_mul_9pass_mt(int fid, size_t n, int party, const u64 key[4], const u64* x, const u64* y, const u64* a,
                           const u64* b, const u64* c, const u64* peer_de, u64* my_de, u64* out, int nthreads) {
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 1024) nthreads = 1024;
    init_fields();
    pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * nthreads);
    mt_job* jobs = (mt_job*)malloc(sizeof(mt_job) * nthreads);
    int rc = 0;
    for (int t = 0; t < nthreads; ++t) {
        mt_job j = {fid, party, n, n * t / nthreads, n * (t + 1) / nthreads, key, x, y, a, b, c, peer_de, my_de, out, 0};
        jobs[t] = j;
        if (nthreads == 1) mt_worker(&jobs[t]);
        else if (pthread_create(&th[t], 0, mt_worker, &jobs[t])) { jobs[t].rc = 2; mt_worker(&jobs[t]); th[t] = 0; }
    }
    for (int t = 0; t < nthreads; ++t) {
        if (nthreads > 1 && jobs[t].rc != 2) pthread_join(th[t], 0);
        if (jobs[t].rc == 1) rc = 1;
    }
    free(th); free(jobs);
    return rc;
}

/* ------------------------------------------------------------------------------------------
 * Curve25519 in twisted-Edwards form (ark_curve25519::EdwardsProjective): -x^2 + y^2 = 1 + d x^2 y^2 over 2^255 - 19,
 * extended coordinates {x, y, t, z}, identity (0, 1, 0, 1).  curve.rs:194-409 delegates to ark-ec; the published
 * formulas are add-2008-hwcd-3 (complete for a = -1) and the affine group law it implements.
 * ---------------------------------------------------------------------------------------- */
#define EQF (ora_get_field(ORA_CURVE25519_FQ))
#define ERF (ora_get_field(ORA_CURVE25519_FR))
static void ed_d2(u64 out[4]) {   /* 2*d, d = -121665/121666 */
    const ora_field* q = EQF;
    u64 a[4] = {121665, 0, 0, 0}, b[4] = {121666, 0, 0, 0}, am[4], bm[4], bi[4], d[4];
    ora_fp_from_canonical(q, a, am); ora_fp_from_canonical(q, b, bm);
    ora_fp_inv(q, bm, bi); ora_fp_mul(q, am, bi, d); ora_fp_neg(q, d, d);
    ora_fp_add(q, d, d, out);
}
void ora_ed_identity(u64 out[16]) {
    const ora_field* q = EQF; memset(out, 0, 128); memcpy(out + 4, q->r, 32); memcpy(out + 12, q->r, 32);
}
/* RFC 8032 base point: y = 4/5, x the even root */
void ora_ed_generator(u64 out[16]) {
    const ora_field* q = EQF;
    static const u64 GX[4] = {0xc9562d608f25d51aULL, 0x692cc7609525a7b2ULL, 0xc0a4e231fdd6dc5cULL, 0x216936d3cd6e53feULL};
    static const u64 GY[4] = {0x6666666666666658ULL, 0x6666666666666666ULL, 0x6666666666666666ULL, 0x6666666666666666ULL};
    ora_fp_from_canonical(q, GX, out); ora_fp_from_canonical(q, GY, out + 4);
    ora_fp_mul(q, out, out + 4, out + 8); memcpy(out + 12, q->r, 32);
}
void ora_ed_neg(const u64 a[16], u64 out[16]) {
    const ora_field* q = EQF; u64 nx[4], nt[4];
    ora_fp_neg(q, a, nx); ora_fp_neg(q, a + 8, nt);
    memcpy(out, nx, 32); memmove(out + 4, a + 4, 32); memcpy(out + 8, nt, 32); memmove(out + 12, a + 12, 32);
}
void ora_ed_add(const u64 p[16], const u64 r[16], u64 out[16]) {
    const ora_field* q = EQF;
    u64 A[4], B[4], C[4], D[4], E[4], F[4], G[4], H[4], t1[4], t2[4], k[4], res[16];
    ed_d2(k);
    ora_fp_sub(q, p + 4, p, t1); ora_fp_sub(q, r + 4, r, t2); ora_fp_mul(q, t1, t2, A);
    ora_fp_add(q, p + 4, p, t1); ora_fp_add(q, r + 4, r, t2); ora_fp_mul(q, t1, t2, B);
    ora_fp_mul(q, p + 8, k, C); ora_fp_mul(q, C, r + 8, C);
    ora_fp_mul(q, p + 12, r + 12, D); ora_fp_add(q, D, D, D);
    ora_fp_sub(q, B, A, E); ora_fp_sub(q, D, C, F); ora_fp_add(q, D, C, G); ora_fp_add(q, B, A, H);
    ora_fp_mul(q, E, F, res); ora_fp_mul(q, G, H, res + 4); ora_fp_mul(q, E, H, res + 8); ora_fp_mul(q, F, G, res + 12);
    memcpy(out, res, 128);
}
/* curve.rs:403-409: [s]P, MSB-first double-and-add through the complete addition law */
void ora_ed_scalar_mul(const u64 pt[16], const u64 scalar_mont[4], u64 out[16]) {
    u64 s[4]; ora_fp_to_canonical(ERF, scalar_mont, s);
    u64 acc[16]; ora_ed_identity(acc);
    for (int i = 255; i >= 0; --i) {
        ora_ed_add(acc, acc, acc);
        if ((s[i / 64] >> (i % 64)) & 1) ora_ed_add(acc, pt, acc);
    }
    memcpy(out, acc, 128);
}
void ora_ed_to_affine(const u64 a[16], u64 out_xy[8]) {
    const ora_field* q = EQF; u64 zi[4];
    ora_fp_inv(q, a + 12, zi); ora_fp_mul(q, a, zi, out_xy); ora_fp_mul(q, a + 4, zi, out_xy + 4);
}
/* ark-serialize compressed TE encoding: y little-endian, bit 7 of the last byte = x > -x */
void ora_ed_to_bytes(const u64 a[16], unsigned char out[32]) {
    const ora_field* q = EQF; u64 xy[8], xc[4], nx[4], nxc[4], yc[4];
    ora_ed_to_affine(a, xy);
    ora_fp_to_canonical(q, xy, xc); ora_fp_neg(q, xy, nx); ora_fp_to_canonical(q, nx, nxc); ora_fp_to_canonical(q, xy + 4, yc);
    for (int i = 0; i < 4; ++i) for (int b = 0; b < 8; ++b) out[8 * i + b] = (unsigned char)(yc[i] >> (8 * b));
    if (geq(xc, nxc) && memcmp(xc, nxc, 32) != 0) out[31] |= 0x80;
}
/* generic square-and-multiply in a field (exponent little-endian u64 limbs) */
static void fp_pow4(const ora_field* f, const u64 base[4], const u64 e[4], u64 out[4]) {
    u64 acc[4]; memcpy(acc, f->r, 32);
    for (int i = 255; i >= 0; --i) { fp_sqr(f, acc, acc); if ((e[i / 64] >> (i % 64)) & 1) ora_fp_mul(f, acc, base, acc); }
    memcpy(out, acc, 32);
}
/* CurvePoint::from_bytes on this curve (curve.rs:110-114 -> ark-ec twisted-Edwards deserialize_compressed with validation):
 * y = the low 255 bits (must be < q), bit 255 = "x is the larger root"; x^2 = (y^2 - 1) / (d y^2 + 1) must be a square
 * (q = 5 mod 8: x = w^((q+3)/8), times sqrt(-1) if x^2 = -w); the point must lie in the prime-order subgroup ([l]P = O,
 * the default is_in_correct_subgroup_assuming_on_curve: cofactor 8).  Returns 1 if valid; else 0 and the identity. */
int ora_ed_from_bytes(const unsigned char in[32], u64 out[16]) {
    const ora_field* q = EQF;
    ora_ed_identity(out);
    const int flag = (in[31] >> 7) & 1;
    u64 yc[4] = {0, 0, 0, 0};
    for (int i = 0; i < 32; ++i) { unsigned char b = in[i]; if (i == 31) b &= 0x7f; yc[i / 8] |= (u64)b << (8 * (i % 8)); }
    if (geq(yc, q->p)) return 0;
    u64 y[4], yy[4], u[4], v[4], d2[4], d[4], two[4] = {2, 0, 0, 0}, twom[4], twoi[4], w[4], vi[4];
    ora_fp_from_canonical(q, yc, y);
    fp_sqr(q, y, yy);
    ora_fp_sub(q, yy, q->r, u);                                    /* y^2 - 1 */
    ed_d2(d2); ora_fp_from_canonical(q, two, twom); ora_fp_inv(q, twom, twoi); ora_fp_mul(q, d2, twoi, d);
    ora_fp_mul(q, d, yy, v); ora_fp_add(q, v, q->r, v);             /* d y^2 + 1 (never zero: -1/d is not a square) */
    ora_fp_inv(q, v, vi); ora_fp_mul(q, u, vi, w);
    u64 e[4], x[4], xx[4], nw[4];
    {   /* (q + 3) / 8 = 2^252 - 2 */
        unsigned __int128 cy = 3;
        for (int i = 0; i < 4; ++i) { cy += q->p[i]; e[i] = (u64)cy; cy >>= 64; }
        for (int i = 0; i < 4; ++i) e[i] = (e[i] >> 3) | (i < 3 ? e[i + 1] << 61 : 0);
    }
    fp_pow4(q, w, e, x);
    fp_sqr(q, x, xx); ora_fp_neg(q, w, nw);
    if (memcmp(xx, w, 32) != 0) {
        if (memcmp(xx, nw, 32) != 0) return 0;                      /* not a square */
        u64 e4[4], s[4];                                            /* sqrt(-1) = 2^((q-1)/4) */
        for (int i = 0; i < 4; ++i) e4[i] = q->p[i];
        e4[0] -= 1;
        for (int i = 0; i < 4; ++i) e4[i] = (e4[i] >> 2) | (i < 3 ? e4[i + 1] << 62 : 0);
        fp_pow4(q, twom, e4, s);
        ora_fp_mul(q, x, s, x);
    }
    u64 nx[4], xc[4], nxc[4];
    ora_fp_neg(q, x, nx); ora_fp_to_canonical(q, x, xc); ora_fp_to_canonical(q, nx, nxc);
    const int x_is_larger = geq(xc, nxc) && memcmp(xc, nxc, 32) != 0;
    u64 pt[16];
    memcpy(pt, (x_is_larger == flag) ? x : nx, 32); memcpy(pt + 4, y, 32); ora_fp_mul(q, pt, y, pt + 8); memcpy(pt + 12, q->r, 32);
    /* subgroup check: [l]P == identity  (x = 0 and y = z) */
    u64 acc[16]; ora_ed_identity(acc);
    const u64* l = ERF->p;
    for (int i = 255; i >= 0; --i) { ora_ed_add(acc, acc, acc); if ((l[i / 64] >> (i % 64)) & 1) ora_ed_add(acc, pt, acc); }
    u64 zero[4] = {0, 0, 0, 0};
    if (memcmp(acc, zero, 32) != 0 || memcmp(acc + 4, acc + 12, 32) != 0) return 0;
    memcpy(out, pt, 128);
    return 1;
}
void ora_ed_batch_add(size_t n, const u64* a, const u64* b, u64* out) { for (size_t i = 0; i < n; ++i) ora_ed_add(a + 16 * i, b + 16 * i, out + 16 * i); }
void ora_ed_batch_neg(size_t n, const u64* a, u64* out) { for (size_t i = 0; i < n; ++i) ora_ed_neg(a + 16 * i, out + 16 * i); }
void ora_ed_batch_scalar_mul(size_t n, const u64* pts, size_t p_div, const u64* scalars, size_t s_div, u64* out) {
    for (size_t i = 0; i < n; ++i) ora_ed_scalar_mul(pts + 16 * (i / p_div), scalars + 4 * (i / s_div), out + 16 * i);
}
void ora_ed_batch_to_affine(size_t n, const u64* pts, u64* out_xy) { for (size_t i = 0; i < n; ++i) ora_ed_to_affine(pts + 16 * i, out_xy + 8 * i); }
/* curve/share.rs:57-60 */
void ora_edshare_batch_add_public(size_t n, int party, const u64 key[4], const u64* shares, const u64* pub, u64* out) {
    for (size_t i = 0; i < n; ++i) {
        u64 kp[16];
        ora_ed_scalar_mul(pub + 16 * i, key, kp);
        if (party == 0) ora_ed_add(shares + 32 * i, pub + 16 * i, out + 32 * i); else memmove(out + 32 * i, shares + 32 * i, 128);
        ora_ed_add(shares + 32 * i + 16, kp, out + 32 * i + 16);
    }
}

/* curve/share.rs:63-65: sub_public = add_public(-rhs) */
void ora_edshare_batch_sub_public(size_t n, int party, const u64 key[4], const u64* shares, const u64* pub, u64* out) {
    for (size_t i = 0; i < n; ++i) { u64 np_[16]; ora_ed_neg(pub + 16 * i, np_); ora_edshare_batch_add_public(1, party, key, shares + 32 * i, np_, out + 32 * i); }
}
void ora_pointshare_batch_sub_public(size_t n, int party, const u64 key[4], const u64* shares, const u64* pub, u64* out) {
    for (size_t i = 0; i < n; ++i) { u64 np_[12]; ora_g1_neg(pub + 12 * i, np_); ora_pointshare_batch_add_public(1, party, key, shares + 24 * i, np_, out + 24 * i); }
}
/* authenticated_curve.rs:215-220: value * mac_key - share.mac(), per element */
void ora_point_mac_check_shares(size_t n, const u64 key[4], const u64* opened, const u64* shares, u64* out) {
    for (size_t i = 0; i < n; ++i) { u64 kv[12], nm[12]; ora_g1_scalar_mul(opened + 12 * i, key, kv); ora_g1_neg(shares + 24 * i + 12, nm); ora_g1_add(kv, nm, out + 12 * i); }
}
void ora_ed_mac_check_shares(size_t n, const u64 key[4], const u64* opened, const u64* shares, u64* out) {
    for (size_t i = 0; i < n; ++i) { u64 kv[16], nm[16]; ora_ed_scalar_mul(opened + 16 * i, key, kv); ora_ed_neg(shares + 32 * i + 16, nm); ora_ed_add(kv, nm, out + 16 * i); }
}
/* curve/share.rs:85-92 / authenticated_curve.rs:796-805: fold with the group law, starting from the identity */
void ora_ed_sum(size_t n, const u64* pts, size_t stride, u64 out[16]) {
    u64 acc[16]; ora_ed_identity(acc);
    for (size_t i = 0; i < n; ++i) { u64 t[16]; ora_ed_add(acc, pts + stride * i, t); memcpy(acc, t, 128); }
    memcpy(out, acc, 128);
}
/* CurvePoint::msm on Curve25519 (curve.rs:549-560, generic over C): the definition, sum_i s_i * P_i */
void ora_ed_msm(size_t n, const u64* pts, const u64* scalars, size_t scalar_stride, u64 out[16]) {
    u64 acc[16], t[16], u[16]; ora_ed_identity(acc);
    for (size_t i = 0; i < n; ++i) { ora_ed_scalar_mul(pts + 16 * i, scalars + scalar_stride * i, t); ora_ed_add(acc, t, u); memcpy(acc, u, 128); }
    memcpy(out, acc, 128);
}
/* CurvePointResult::msm_authenticated (curve.rs:618-642): PointShare(msm(shares, points), msm(macs, points)) */
void ora_ed_msm_authenticated(size_t n, const u64* pts, const u64* scalar_shares, u64 out[32]) {
    ora_ed_msm(n, pts, scalar_shares, 8, out);
    ora_ed_msm(n, pts, scalar_shares + 4, 8, out + 16);
}
/* authenticated_curve.rs:127-131: my + peer == identity; returns 1 when it is.  Compared on affine coordinates (0, 1). */
int ora_ed_is_identity_sum(const u64 a[16], const u64 b[16]) {
    u64 s[16], xy[8], one[4];
    ora_ed_add(a, b, s); ora_ed_to_affine(s, xy);
    memcpy(one, ora_get_field(ORA_CURVE25519_FQ)->r, 32);
    return is_zero4(xy) && memcmp(xy + 4, one, 32) == 0;
}

/* ------------------------------------------------------------------------------------------
 * Range-parallel forms of batch functions above, for the full-size parity tests (BASELINE sizes: 2^24 shares,
 * 2^18 PointShare x Scalar): the SAME per-element functions, a static contiguous range split over pthreads.
 * Elements are independent in the reference as well (authenticated_scalar.rs:299-311, curve/share.rs:108-114).
 * ---------------------------------------------------------------------------------------- */
typedef void (*range_fn)(void* ctx, size_t lo, size_t hi);
typedef struct { range_fn fn; void* ctx; size_t lo, hi; } range_job;
static void* range_worker(void* arg) { range_job* j = (range_job*)arg; if (j->hi > j->lo) j->fn(j->ctx, j->lo, j->hi); return 0; }
static void par_for(size_t n, int nthreads, range_fn fn, void* ctx) {
    init_fields();
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 1024) nthreads = 1024;
    if ((size_t)nthreads > n) nthreads = n ? (int)n : 1;
    if (nthreads == 1) { if (n) fn(ctx, 0, n); return; }
    pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * nthreads);
    range_job* jobs = (range_job*)malloc(sizeof(range_job) * nthreads);
    char* started = (char*)calloc(nthreads, 1);
    for (int t = 0; t < nthreads; ++t) {
        range_job j = {fn, ctx, n * t / nthreads, n * (t + 1) / nthreads};
        jobs[t] = j;
        if (pthread_create(&th[t], 0, range_worker, &jobs[t]) == 0) started[t] = 1;
        else range_worker(&jobs[t]);          /* could not spawn: run the range inline */
    }
    for (int t = 0; t < nthreads; ++t) if (started[t]) pthread_join(th[t], 0);
    free(th); free(jobs); free(started);
}

typedef struct { int fid; const u64 *key, *shares, *peer; u64 *opened, *chk; } omc_ctx;
static void omc_range(void* p, size_t lo, size_t hi) {
    omc_ctx* c = (omc_ctx*)p;
    const ora_field* f = ora_get_field(c->fid);
    for (size_t i = lo; i < hi; ++i)                                  /* open_batch combine gate :161-171 on the `.share()` halves */
        ora_fp_add(f, c->shares + 8 * i, c->peer + 4 * i, c->opened + 4 * i);
    ora_mac_check_shares(c->fid, hi - lo, c->key, c->opened + 4 * lo, c->shares + 8 * lo, c->chk + 4 * lo);   /* :299-311 */
}
/* open_authenticated_batch, one party's local work (authenticated_scalar.rs:278-311): opened_i = share_i + peer_i,
 * chk_i = mac_key * opened_i - mac_i */
void ora_open_and_mac_check_mt(int fid, size_t n, const u64 key[4], const u64* shares, const u64* peer, u64* out_opened,
                               u64* out_chk, int nthreads) {
    omc_ctx c = {fid, key, shares, peer, out_opened, out_chk};
    par_for(n, nthreads, omc_range, &c);
}

typedef struct { const u64 *shares, *scalars; u64* out; } psm_ctx;
static void psm_range(void* p, size_t lo, size_t hi) {
    psm_ctx* c = (psm_ctx*)p;
    ora_pointshare_batch_mul_public(hi - lo, c->shares + 24 * lo, c->scalars + 4 * lo, c->out + 24 * lo);
}
void ora_pointshare_batch_mul_public_mt(size_t n, const u64* shares, const u64* scalars, u64* out, int nthreads) {
    psm_ctx c = {shares, scalars, out};
    par_for(n, nthreads, psm_range, &c);
}

typedef struct { const u64* pts; u64* xy; unsigned char* inf; } aff_ctx;
static void aff_range(void* p, size_t lo, size_t hi) {
    aff_ctx* c = (aff_ctx*)p;
    ora_g1_batch_to_affine(hi - lo, c->pts + 12 * lo, c->xy + 8 * lo, c->inf + lo);
}
void ora_g1_batch_to_affine_mt(size_t n, const u64* pts, u64* out_xy, unsigned char* is_inf, int nthreads) {
    aff_ctx c = {pts, out_xy, is_inf};
    par_for(n, nthreads, aff_range, &c);
}

typedef struct { int fid; size_t n; const u64 *x, *y, *a, *b; u64* de; } bmk_ctx;
static void bmk_range(void* p, size_t lo, size_t hi) {
    bmk_ctx* c = (bmk_ctx*)p;
    const ora_field* f = ora_get_field(c->fid);
    for (size_t i = lo; i < hi; ++i) {                                /* ora_beaver_mask on a range of the full d||e buffer */
        ora_fp_sub(f, c->x + 8 * i, c->a + 8 * i, c->de + 4 * i);
        ora_fp_sub(f, c->y + 8 * i, c->b + 8 * i, c->de + 4 * (c->n + i));
    }
}
void ora_beaver_mask_mt(int fid, size_t n, const u64* x, const u64* y, const u64* a, const u64* b, u64* out_de, int nthreads) {
    bmk_ctx c = {fid, n, x, y, a, b, out_de};
    par_for(n, nthreads, bmk_range, &c);
}

typedef struct { const u64 *pts, *scalars; size_t p_div, s_div; u64* out; } edm_ctx;
static void edm_range(void* p, size_t lo, size_t hi) {
    edm_ctx* c = (edm_ctx*)p;
    for (size_t i = lo; i < hi; ++i) ora_ed_scalar_mul(c->pts + 16 * (i / c->p_div), c->scalars + 4 * (i / c->s_div), c->out + 16 * i);
}
void ora_ed_batch_scalar_mul_mt(size_t n, const u64* pts, size_t p_div, const u64* scalars, size_t s_div, u64* out, int nthreads) {
    edm_ctx c = {pts, scalars, p_div, s_div, out};
    par_for(n, nthreads, edm_range, &c);
}
typedef struct { const u64* pts; u64* xy; } eda_ctx;
static void eda_range(void* p, size_t lo, size_t hi) {
    eda_ctx* c = (eda_ctx*)p;
    ora_ed_batch_to_affine(hi - lo, c->pts + 16 * lo, c->xy + 8 * lo);
}
void ora_ed_batch_to_affine_mt(size_t n, const u64* pts, u64* out_xy, int nthreads) {
    eda_ctx c = {pts, out_xy};
    par_for(n, nthreads, eda_range, &c);
}

/* The FUSED single-pass form of one party's batch_mul: per gate, what the single-gate `Mul` does (authenticated_scalar.rs:799-843) -- the
 * two masking subtractions (:809-812; only the `.share()` halves are ever used, :141-145), the combine with the peer's payload (:161-171) and
 * the gate closure  res = d * b_share + e * a_share + c_share;  res.add_public(de, mac_key, party_id)  (:835-840) -- in ONE sweep over
 * memory instead of the nine of ora_batch_mul_9pass_local.  BASELINE.md section 3 asks for this beside the 9-pass "so that the comparison
 * is not unfairly hobbled by pass count".  Same words as the 9-pass (field addition is associative and commutative on canonical residues;
 * tests/test_oracle_field.py checks it).  peer_de / my_de are d||e buffers of the FULL batch. */
typedef struct { int fid, party; size_t n; const u64 *key, *x, *y, *a, *b, *c, *peer_de; u64 *my_de, *out; } fus_ctx;
static void fus_range(void* p, size_t lo, size_t hi) {
    fus_ctx* q = (fus_ctx*)p;
    const ora_field* f = ora_get_field(q->fid);
    const size_t n = q->n;
    for (size_t i = lo; i < hi; ++i) {
        u64 d[4], e[4], de[4], db[8], ea[8], t[8], res[8];
        ora_fp_sub(f, q->x + 8 * i, q->a + 8 * i, q->my_de + 4 * i);            /* masked_lhs.share()                 :809-812, :141-145 */
        ora_fp_sub(f, q->y + 8 * i, q->b + 8 * i, q->my_de + 4 * (n + i));      /* masked_rhs.share()                                    */
        ora_fp_add(f, q->my_de + 4 * i, q->peer_de + 4 * i, d);                 /* open: mine + peer                  :161-171           */
        ora_fp_add(f, q->my_de + 4 * (n + i), q->peer_de + 4 * (n + i), e);
        ora_fp_mul(f, d, e, de);                                                /* let de = d * e                     :836               */
        share_mul_public(f, q->b + 8 * i, d, db);                               /* d * b_share                        :837               */
        share_mul_public(f, q->a + 8 * i, e, ea);                               /* e * a_share                                           */
        share_add(f, db, ea, t);
        share_add(f, t, q->c + 8 * i, res);                                     /* + c_share                                             */
        share_add_public(f, q->party, q->key, res, de, q->out + 8 * i);         /* res.add_public(de, mac_key, party) :838               */
    }
}
int ora_batch_mul_fused_mt(int fid, size_t n, int party, const u64 key[4], const u64* x, const u64* y, const u64* a, const u64* b,
                           const u64* c, const u64* peer_de, u64* my_de, u64* out, int nthreads) {
    fus_ctx q = {fid, party, n, key, x, y, a, b, c, peer_de, my_de, out};
    par_for(n, nthreads, fus_range, &q);
    return 0;
}
