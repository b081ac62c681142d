"""GPU parity for the Curve25519 (twisted Edwards) kernels vs the CPU oracle, on affine coordinates."""
import numpy as np
import pytest

import pyref
from helpers import mont_array, rand_values, limbs_to_ints

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng(pkg):
    return pkg.Engine("curve25519_fr", device=0, host_buffers=True)


def ext(points, zs):
    return np.array(sum((pyref.ed_extended_mont(p, z) for p, z in zip(points, zs)), []), dtype=np.uint64)


def rand_points(n, seed):
    pts = [pyref.ed_mul(pyref.ED_B, k) for k in rand_values(2, n, seed)]
    if n >= 3:
        pts[1] = (0, 1)
    return pts, ext(pts, [1 + (seed + 5 * i) % 97 for i in range(n)])


def aff_equal(eng, oracle, a, b):
    n = len(a) // 16
    xa = np.zeros(8 * n, dtype=np.uint64); eng.ed_to_affine(n, np.ascontiguousarray(a), xa)
    return np.array_equal(xa, oracle.ed_batch_to_affine(np.ascontiguousarray(b)))


def test_add_sub_neg_and_encoding(eng, oracle):
    n = 50
    pts, P = rand_points(n, 1)
    _, Q = rand_points(n, 2)
    Q[16 * 4:16 * 5] = P[16 * 4:16 * 5]                      # P + P through the unified addition
    o = np.zeros(16 * n, dtype=np.uint64); eng.ed_add(n, P, Q, o)
    assert aff_equal(eng, oracle, o, oracle.ed_batch_add(P, Q))
    eng.ed_sub(n, P, Q, o)
    assert aff_equal(eng, oracle, o, oracle.ed_batch_add(P, oracle.ed_batch_neg(Q)))
    eng.ed_neg(n, P, o)
    assert aff_equal(eng, oracle, o, oracle.ed_batch_neg(P))
    b = np.zeros(32 * n, dtype=np.uint8); eng.ed_to_bytes(n, P, b)
    assert b.tobytes() == b"".join(pyref.ed_compress(p) for p in pts)


def test_scalar_mul_and_generator(eng, oracle):
    n = 40
    pts, P = rand_points(n, 3)
    ks = [0, 1, 2, pyref.EL - 1, 15, 16] + rand_values(2, n - 6, 4)
    S = mont_array(2, ks)
    o = np.zeros(16 * n, dtype=np.uint64); eng.ed_scalar_mul(n, P, S, o)
    assert aff_equal(eng, oracle, o, oracle.ed_batch_scalar_mul(P, S))
    xa = np.zeros(8 * n, dtype=np.uint64); eng.ed_to_affine(n, o, xa)
    for i in range(n):
        x, y = limbs_to_ints(xa[8 * i:8 * i + 8])
        assert (pyref.from_mont(4, x), pyref.from_mont(4, y)) == pyref.ed_mul(pts[i], ks[i])
    eng.ed_generator_mul(n, S, o)
    G = ext([pyref.ED_B] * n, [1] * n)
    assert aff_equal(eng, oracle, o, oracle.ed_batch_scalar_mul(G, S))


@pytest.mark.parametrize("party", [0, 1])
def test_edshare_ops(eng, oracle, party):
    n = 8
    _, A = rand_points(2 * n, 5)
    _, B = rand_points(2 * n, 6)
    _, PUB = rand_points(n, 7)
    S = mont_array(2, [0, 1] + rand_values(2, n - 2, 8))
    key = mont_array(2, rand_values(2, 1, 9))
    SS = mont_array(2, rand_values(2, 2 * n, 10))
    o = np.zeros(32 * n, dtype=np.uint64)
    eng.edshare_add(n, A, B, o); assert aff_equal(eng, oracle, o, oracle.ed_batch_add(A, B))
    eng.edshare_sub(n, A, B, o); assert aff_equal(eng, oracle, o, oracle.ed_batch_add(A, oracle.ed_batch_neg(B)))
    eng.edshare_neg(n, A, o); assert aff_equal(eng, oracle, o, oracle.ed_batch_neg(A))
    eng.edshare_mul_public(n, A, S, o)                         # (share * s, mac * s): point j uses scalar j // 2
    assert aff_equal(eng, oracle, o, oracle.ed_batch_scalar_mul(A, S, n=2 * n, s_div=2))
    eng.edshare_add_public(n, party, key, A, PUB, o)
    assert aff_equal(eng, oracle, o, oracle.edshare_add_public(party, key, A, PUB))
    eng.scalarshare_mul_ed_generator(n, SS, o)
    G = ext([pyref.ED_B], [1])
    assert aff_equal(eng, oracle, o, oracle.ed_batch_scalar_mul(G, SS, n=2 * n, p_div=2 * n))


def test_wrong_context_is_rejected(pkg):
    e = pkg.Engine("bn254_fr", device=0, host_buffers=True)
    with pytest.raises(pkg.ArkMpcError):
        e.ed_neg(1, np.zeros(16, dtype=np.uint64), np.zeros(16, dtype=np.uint64))
    e.close()


def test_from_bytes(eng, oracle):
    """arkmpc_ed_from_bytes vs the oracle: round trip of to_bytes (exact limbs: no addition chain involved) and rejection of
    a non-canonical y, a non-residue, a point of order 2 and points outside the prime-order subgroup."""
    n = 120
    pts, P = rand_points(n, 51)
    data = np.zeros(32 * n, dtype=np.uint8); eng.ed_to_bytes(n, P, data)
    bad = [int(pyref.EQ).to_bytes(32, "little"), int(pyref.EQ - 1).to_bytes(32, "little"), (2).to_bytes(32, "little")]
    bad += [int(y).to_bytes(32, "little") for y in range(3, 40) if pyref.ed_decompress(int(y).to_bytes(32, "little")) is None]
    data = np.concatenate([data, np.frombuffer(b"".join(bad), dtype=np.uint8)])
    m = len(data) // 32
    out = np.zeros(16 * m, dtype=np.uint64); ok = np.zeros(m, dtype=np.uint8)
    eng.ed_from_bytes(m, data, out, ok)
    want, want_ok = oracle.ed_from_bytes(data)
    assert np.array_equal(ok, want_ok) and np.array_equal(out, want)
    assert ok[:n].all() and not ok[n:].any()
    assert aff_equal(eng, oracle, out[:16 * n], P)
