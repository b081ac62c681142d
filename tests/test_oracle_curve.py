"""Pins the oracle's BN254 G1 arithmetic against the affine group law in Python ints and against the
EIP-196 (alt_bn128) published points.  CPU only."""
import numpy as np
import pytest

import pyref
from helpers import mont_array, limbs_to_ints, rand_values

# EIP-196 / go-ethereum bn256 test vector: 2 * (1, 2) on y^2 = x^3 + 3 over Fq
TWO_G = (0x030644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd3,
         0x15ed738c0e0a7c92e7845f96b2ae9c0a68a6a449e3538fc7ff3ebf7a5a18a2c4)


def affine_of(oracle, pts):
    xy, inf = oracle.g1_batch_to_affine(pts)
    out = []
    for i in range(len(inf)):
        if inf[i]:
            out.append(None)
        else:
            x, y = limbs_to_ints(xy[8 * i:8 * i + 8])
            out.append((pyref.from_mont(3, x), pyref.from_mont(3, y)))
    return out


def jac(points, zs=None):
    zs = zs or [1] * len(points)
    return np.array(sum((pyref.g1_jacobian_mont(p, z) for p, z in zip(points, zs)), []), dtype=np.uint64)


def test_python_group_law_matches_eip196():
    assert pyref.g1_add(pyref.G, pyref.G) == TWO_G
    assert pyref.g1_mul(pyref.G, 2) == TWO_G
    assert pyref.g1_mul(pyref.G, pyref.RORD) is None          # group order
    assert (TWO_G[1] ** 2 - TWO_G[0] ** 3 - 3) % pyref.Q == 0


def test_generator_identity_and_double(oracle):
    g = oracle.g1_generator()
    assert affine_of(oracle, g) == [pyref.G]
    assert affine_of(oracle, oracle.g1_identity()) == [None]
    assert affine_of(oracle, oracle.g1_batch_add(g, g)) == [TWO_G]          # P + P through the add path
    assert affine_of(oracle, oracle.g1_batch_add(g, oracle.g1_neg(g))) == [None]   # P + (-P)
    assert affine_of(oracle, oracle.g1_batch_add(g, oracle.g1_identity())) == [pyref.G]
    assert affine_of(oracle, oracle.g1_batch_add(oracle.g1_identity(), g)) == [pyref.G]


def test_add_and_scalar_mul_vs_python(oracle):
    ks = [0, 1, 2, 3, pyref.RORD - 1, pyref.RORD - 2, (1 << 253) % pyref.RORD] + rand_values(0, 9, 42)
    base_k = rand_values(0, len(ks), 43)
    pts = [pyref.g1_mul(pyref.G, k) for k in base_k]
    zs = [1 + (i * 7919) % 1000 for i in range(len(ks))]               # non-trivial Jacobian representatives
    P = jac(pts, zs)
    got = affine_of(oracle, oracle.g1_batch_scalar_mul(P, mont_array(0, ks)))
    assert got == [pyref.g1_mul(p, k) for p, k in zip(pts, ks)]
    Q = jac(list(reversed(pts)))
    assert affine_of(oracle, oracle.g1_batch_add(P, Q)) == [pyref.g1_add(a, b) for a, b in zip(pts, reversed(pts))]
    # compressed encoding
    assert oracle.g1_to_bytes(P).tobytes() == b"".join(pyref.g1_compress(p) for p in pts)
    assert oracle.g1_to_bytes(oracle.g1_identity()).tobytes() == pyref.g1_compress(None)


def test_msm_vs_python(oracle):
    """CurvePoint::msm / msm_authenticated (curve.rs:549-560, 618-642) against the Python affine group law."""
    n = 12
    ks = [0, 1, pyref.RORD - 1] + rand_values(0, n - 3, 77)
    macs = rand_values(0, n, 78)
    base_k = rand_values(0, n, 79)
    pts = [pyref.g1_mul(pyref.G, k) for k in base_k]
    pts[4] = None                                                         # identity among the bases
    pts[6] = pts[5]                                                       # repeated base
    P = jac([p if p else (1, 1) for p in pts], [0 if p is None else 1 + 31 * i for i, p in enumerate(pts)])
    def ref(scalars):
        acc = None
        for k, p in zip(scalars, pts):
            acc = pyref.g1_add(acc, pyref.g1_mul(p, k) if p else None)
        return acc
    assert affine_of(oracle, oracle.g1_msm(P, mont_array(0, ks))) == [ref(ks)]
    shares = mont_array(0, [v for pair in zip(ks, macs) for v in pair])
    got = oracle.g1_msm_authenticated(P, shares)
    assert affine_of(oracle, got) == [ref(ks), ref(macs)]
    assert affine_of(oracle, oracle.g1_msm(P[:0], mont_array(0, []))) == [None]     # empty -> identity


def test_from_bytes_vs_python(oracle):
    """CurvePoint::from_bytes (curve.rs:110-114): inverse of the compressed encoding, with arkworks' validation."""
    ks = [1, 2, 3, pyref.RORD - 1] + rand_values(0, 20, 90)
    pts = [pyref.g1_mul(pyref.G, k) for k in ks] + [None]
    data = np.frombuffer(b"".join(pyref.g1_compress(p) for p in pts), dtype=np.uint8).copy()
    out, ok = oracle.g1_from_bytes(data)
    assert ok.tolist() == [1] * len(pts)
    assert affine_of(oracle, out) == pts
    assert np.array_equal(out[-12:], oracle.g1_identity())
    for i in range(len(ks)):                                              # decompressed points are (x, y, 1)
        assert np.array_equal(out[12 * i + 8:12 * i + 12], oracle.g1_identity()[:4])
    # invalid encodings: x >= q, both flags, x^3 + 3 a non-residue
    bad = []
    b = bytearray(int(pyref.Q).to_bytes(32, "little")); bad.append(bytes(b))
    b = bytearray(pyref.g1_compress(pts[0])); b[31] |= 0xC0; bad.append(bytes(b))
    x = 0
    while pow((x ** 3 + 3) % pyref.Q, (pyref.Q - 1) // 2, pyref.Q) == 1 or (x ** 3 + 3) % pyref.Q == 0:
        x += 1
    bad.append(int(x).to_bytes(32, "little"))
    out, ok = oracle.g1_from_bytes(np.frombuffer(b"".join(bad), dtype=np.uint8).copy())
    assert ok.tolist() == [0, 0, 0]
