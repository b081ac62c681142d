"""Pins the C oracle (oracle/ark_oracle.c) against exact big-int arithmetic and published constants.
CPU only.  The reference holds no golden vectors for this path, so these are
the anchors: Python ints, arkworks' published R/INV, hashlib SHA3."""
import ctypes
import hashlib

import numpy as np
import pytest

import pyref
from helpers import mont_array, from_mont_array, mixed_values, limbs_to_ints, ints_to_limbs

FIDS = [0, 1, 2, 3]


class OraField(ctypes.Structure):
    _fields_ = [("p", ctypes.c_uint64 * 4), ("r", ctypes.c_uint64 * 4), ("r2", ctypes.c_uint64 * 4),
                ("inv", ctypes.c_uint64), ("bits", ctypes.c_int)]


@pytest.mark.parametrize("fid", FIDS)
def test_montgomery_constants_match_published(oracle, fid):
    f = ctypes.cast(oracle.lib.ora_get_field(ctypes.c_int(fid)), ctypes.POINTER(OraField)).contents
    p = pyref.unlimbs(f.p)
    assert p == pyref.P[fid]
    assert pyref.unlimbs(f.r) == (1 << 256) % p == pyref.PUBLISHED_R[fid]
    assert pyref.unlimbs(f.r2) == (1 << 512) % p
    assert f.inv == (-pow(p, -1, 1 << 64)) % (1 << 64) == pyref.PUBLISHED_INV[fid]
    assert f.bits == p.bit_length()


@pytest.mark.parametrize("fid", FIDS + [4])
def test_field_ops_vs_bigint(oracle, fid):
    p = pyref.P[fid]
    a = mixed_values(fid, 200, seed=100 + fid)
    b = list(reversed(mixed_values(fid, 200, seed=200 + fid)))
    am, bm = mont_array(fid, a), mont_array(fid, b)
    assert from_mont_array(fid, oracle.scalar_add(fid, am, bm)) == [(x + y) % p for x, y in zip(a, b)]
    assert from_mont_array(fid, oracle.scalar_sub(fid, am, bm)) == [(x - y) % p for x, y in zip(a, b)]
    assert from_mont_array(fid, oracle.scalar_mul(fid, am, bm)) == [(x * y) % p for x, y in zip(a, b)]
    assert from_mont_array(fid, oracle.scalar_neg(fid, am)) == [(-x) % p for x in a]
    # conversions: canonical <-> Montgomery, including inputs >= p that must be reduced first
    raw = a + [p, p + 5, (1 << 256) - 1]
    assert limbs_to_ints(oracle.from_canonical(fid, ints_to_limbs(raw))) == [pyref.to_mont(fid, v) for v in raw]
    assert limbs_to_ints(oracle.to_canonical(fid, am)) == a
    # outputs are canonical residues (< p) in Montgomery form
    assert all(v < p for v in limbs_to_ints(oracle.scalar_mul(fid, am, bm)))


@pytest.mark.parametrize("fid", FIDS)
def test_inverse_and_bytes(oracle, fid):
    p = pyref.P[fid]
    vals = [v for v in mixed_values(fid, 40, seed=7) if v != 0]
    f = ctypes.c_void_p(oracle.lib.ora_get_field(ctypes.c_int(fid)))
    for v in vals[:12]:
        inp = ints_to_limbs([pyref.to_mont(fid, v)])
        out = np.zeros(4, dtype=np.uint64)
        oracle.lib.ora_fp_inv(f, oracle._p(inp), oracle._p(out))
        assert pyref.from_mont(fid, limbs_to_ints(out)[0]) == pow(v, -1, p)
    be = oracle.to_bytes_be(fid, mont_array(fid, vals)).tobytes()
    assert be == b"".join(pyref.to_bytes_be(fid, v) for v in vals)
    # from_be_bytes_mod_order on 32-byte digests and on longer strings
    for data in [b"\xff" * 32, bytes(range(32)), b"\x01" + b"\x00" * 40, b""]:
        buf = np.frombuffer(data if data else b"\0", dtype=np.uint8).copy()
        out = np.zeros(4, dtype=np.uint64)
        oracle.lib.ora_fp_from_be_bytes_mod_order(f, oracle._p(buf), ctypes.c_size_t(len(data)), oracle._p(out))
        assert pyref.from_mont(fid, limbs_to_ints(out)[0]) == int.from_bytes(data, "big") % p


def test_sha3_known_answers(oracle):
    # FIPS 202 known answers + hashlib over the padding boundaries (rate = 136 bytes)
    assert oracle.sha3_256(b"").hex() == "a7ffc6f8bf1ed76651c14756a061d662f580ff4de43b49fa82d80a4b80f8434a"
    assert oracle.sha3_256(b"abc").hex() == "3a985da74fe225b2045c172d6bd390bd855f086e3e9d525b46bfe24511431532"
    for n in [1, 55, 135, 136, 137, 271, 272, 273, 1000, 4096 + 17]:
        data = bytes((i * 131 + 7) & 0xFF for i in range(n))
        assert oracle.sha3_256(data) == hashlib.sha3_256(data).digest()


@pytest.mark.parametrize("fid", [0, 1, 2])
def test_commitment_vs_hashlib(oracle, fid):
    vals = mixed_values(fid, 37, seed=5)
    blinder = 0x1234567890abcdef1234567890abcdef % pyref.P[fid]
    got = oracle.commit_scalars(fid, mont_array(fid, vals), mont_array(fid, [blinder]))
    assert pyref.from_mont(fid, limbs_to_ints(got)[0]) == pyref.commit(fid, vals, blinder)


@pytest.mark.parametrize("fid", FIDS)
def test_batch_inverse_vs_bigint(oracle, fid):
    p = pyref.P[fid]
    vals = mixed_values(fid, 50, seed=9)            # includes 0, which must stay 0 (ark_ff::batch_inversion)
    got = from_mont_array(fid, oracle.scalar_batch_inverse(fid, mont_array(fid, vals)))
    assert got == [0 if v == 0 else pow(v, -1, p) for v in vals]


# ark-bn254 / ark-bls12-381 `FrConfig`: TWO_ADICITY and TWO_ADIC_ROOT_OF_UNITY (published constants of the arkworks crates)
TWO_ADIC = {
    0: (28, 19103219067921713944291392827692070036145651957329286315305642004821462161904),
    1: (32, 10238227357739495823651030575849232062558860180284477541189508159991286009131),
}


@pytest.mark.parametrize("fid", [0, 1])
def test_two_adic_root_of_unity_through_oracle_mul(oracle, fid):
    """root^(2^s) == 1 and root^(2^(s-1)) == -1, computed ONLY with the oracle's Montgomery multiplication."""
    s, root = TWO_ADIC[fid]
    p = pyref.P[fid]
    x = mont_array(fid, [root])
    for i in range(s - 1):
        x = oracle.scalar_mul(fid, x, x)
    assert from_mont_array(fid, x) == [p - 1]
    x = oracle.scalar_mul(fid, x, x)
    assert from_mont_array(fid, x) == [1]


def test_prefix_product_vs_bigint(oracle):
    fid = 0; p = pyref.P[fid]
    vals = mixed_values(fid, 30, seed=4)[3:]
    got = from_mont_array(fid, oracle.scalar_prefix_product(fid, mont_array(fid, vals)))
    run, want = 1, []
    for v in vals:
        run = run * v % p; want.append(run)
    assert got == want


@pytest.mark.parametrize("fid", FIDS)
def test_reductions_vs_bigint(oracle, fid):
    """Sum / Product for ScalarResult and Sum for ScalarShare (scalar_result.rs:325-338, share.rs:103-111) against Python ints,
    including the empty fold (0 / 1)."""
    import functools
    p = pyref.P[fid]
    for n in (0, 1, 2, 257):
        a = mixed_values(fid, n, seed=900 + n + fid)
        am = mont_array(fid, a)
        assert from_mont_array(fid, oracle.scalar_sum(fid, am)) == [sum(a) % p]
        assert from_mont_array(fid, oracle.scalar_product(fid, am)) == [functools.reduce(lambda x, y: x * y % p, a, 1)]
        b = mixed_values(fid, n, seed=901 + n + fid)[::-1]
        rec = np.ascontiguousarray(np.concatenate([am.reshape(-1, 4), mont_array(fid, b).reshape(-1, 4)], axis=1).reshape(-1))
        assert from_mont_array(fid, oracle.share_sum(fid, rec)) == [sum(a) % p, sum(b) % p]


@pytest.mark.parametrize("fid", [0, 1, 2])
@pytest.mark.parametrize("threads", [1, 3])
def test_fused_single_pass_batch_mul_equals_the_nine_passes_and_bigints(oracle, fid, threads):
    """BASELINE.md section 3 times two CPU forms of batch_mul: the reference's literal nine passes (authenticated_scalar.rs:848-879) and the
    fused single pass (the single-gate Mul's closure, :799-843).  Both parties through both forms: identical words; and the opened product is
    x * y with MAC key * x * y in Python ints."""
    from helpers import authenticated_shares, rand_values
    p = pyref.P[fid]
    n = 333
    k0, k1 = rand_values(fid, 2, 41 + fid)
    key = (k0 + k1) % p
    keys = [mont_array(fid, [k0]), mont_array(fid, [k1])]
    x, y, a, b = (mixed_values(fid, n, seed=50 + i + fid) if i < 2 else rand_values(fid, n, 60 + i + fid) for i in range(4))
    c = [(u * v) % p for u, v in zip(a, b)]
    sh = {nm: authenticated_shares(fid, v, key, 70 + i) for i, (nm, v) in enumerate(zip("xyabc", (x, y, a, b, c)))}
    de = [oracle.beaver_mask(fid, sh["x"][q], sh["y"][q], sh["a"][q], sh["b"][q]) for q in (0, 1)]
    outs = []
    for q in (0, 1):
        args = (fid, q, keys[q], sh["x"][q], sh["y"][q], sh["a"][q], sh["b"][q], sh["c"][q], de[1 - q])
        d9, o9 = oracle.batch_mul_9pass_local(*args)
        df, of = oracle.batch_mul_fused_mt(*args, nthreads=threads)
        assert np.array_equal(d9, de[q]) and np.array_equal(df, de[q])
        assert np.array_equal(o9, of)
        outs.append(of.reshape(n, 8))
    share = [(u + v) % p for u, v in zip(from_mont_array(fid, np.ascontiguousarray(outs[0][:, :4]).reshape(-1)), from_mont_array(fid, np.ascontiguousarray(outs[1][:, :4]).reshape(-1)))]
    mac = [(u + v) % p for u, v in zip(from_mont_array(fid, np.ascontiguousarray(outs[0][:, 4:]).reshape(-1)), from_mont_array(fid, np.ascontiguousarray(outs[1][:, 4:]).reshape(-1)))]
    assert share == [(u * v) % p for u, v in zip(x, y)]
    assert mac == [(key * u * v) % p for u, v in zip(x, y)]
