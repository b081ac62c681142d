"""Pins the oracle's Curve25519 (twisted Edwards) arithmetic: RFC 8032's base point and test-vector public key, the group
order, and the affine Edwards addition law in Python ints.  CPU only."""
import hashlib

import numpy as np

import pyref
from helpers import mont_array, limbs_to_ints, rand_values


def affine(oracle, pts):
    xy = oracle.ed_batch_to_affine(pts)
    out = []
    for i in range(len(pts) // 16):
        x, y = limbs_to_ints(xy[8 * i:8 * i + 8])
        out.append((pyref.from_mont(4, x), pyref.from_mont(4, y)))
    return out


def ext(points, zs=None):
    zs = zs or [1] * len(points)
    return np.array(sum((pyref.ed_extended_mont(p, z) for p, z in zip(points, zs)), []), dtype=np.uint64)


def test_base_point_order_and_curve_equation(oracle):
    B = pyref.ED_B
    assert (-B[0] ** 2 + B[1] ** 2 - 1 - pyref.ED_D * B[0] ** 2 * B[1] ** 2) % pyref.EQ == 0
    assert pyref.ed_mul(B, pyref.EL) == (0, 1)
    assert affine(oracle, oracle.ed_generator()) == [B]
    assert affine(oracle, oracle.ed_identity()) == [(0, 1)]


def test_rfc8032_public_key_vector(oracle):
    """RFC 8032 section 7.1 TEST 1: A = [s]B with s the clamped low half of SHA-512(secret key)."""
    sk = bytes.fromhex("9d61b19deffd5a60ba844af492ec2cc44449c5697b326919703bac031cae7f60")
    pk = bytes.fromhex("d75a980182b10ab7d54bfed3c964073a0ee172f3daa62325af021a68f707511a")
    h = bytearray(hashlib.sha512(sk).digest()[:32])
    h[0] &= 248; h[31] &= 127; h[31] |= 64
    s = int.from_bytes(h, "little") % pyref.EL
    got = affine(oracle, oracle.ed_batch_scalar_mul(oracle.ed_generator(), mont_array(2, [s])))[0]
    y = int.from_bytes(pk, "little") & ((1 << 255) - 1)
    assert got[1] == y and (got[0] & 1) == pk[31] >> 7          # RFC 8032 encodes y and the parity of x
    assert got == pyref.ed_mul(pyref.ED_B, s)


def test_group_law_vs_python(oracle):
    ks = [0, 1, 2, pyref.EL - 1, 8, (1 << 251) + 5] + rand_values(2, 10, 5)
    base = rand_values(2, len(ks), 6)
    pts = [pyref.ed_mul(pyref.ED_B, b) for b in base]
    pts[2] = (0, 1)                                              # identity as an operand
    zs = [1 + 13 * i for i in range(len(ks))]
    P = ext(pts, zs)
    assert affine(oracle, P) == pts
    assert affine(oracle, oracle.ed_batch_scalar_mul(P, mont_array(2, ks))) == [pyref.ed_mul(p, k) for p, k in zip(pts, ks)]
    Q = ext(list(reversed(pts)))
    assert affine(oracle, oracle.ed_batch_add(P, Q)) == [pyref.ed_add(a, b) for a, b in zip(pts, reversed(pts))]
    assert affine(oracle, oracle.ed_batch_add(P, ext(pts))) == [pyref.ed_add(a, a) for a in pts]           # doubling via add
    assert affine(oracle, oracle.ed_batch_add(P, oracle.ed_batch_neg(ext(pts)))) == [(0, 1)] * len(pts)       # P + (-P)
    assert oracle.ed_to_bytes(P).tobytes() == b"".join(pyref.ed_compress(p) for p in pts)


def test_from_bytes_vs_python(oracle):
    """CurvePoint::from_bytes on Curve25519: decompression with arkworks' checks (canonical y, square, prime-order subgroup)."""
    ks = [1, 2, 3, pyref.EL - 1] + rand_values(2, 12, 91)
    pts = [pyref.ed_mul(pyref.ED_B, k) for k in ks] + [(0, 1)]
    good = [pyref.ed_compress(p) for p in pts]
    # not encodings: y >= q; a y whose x^2 is a non-residue; a point of small order (y = -1: order 2); a point outside the subgroup
    bad = [int(pyref.EQ).to_bytes(32, "little"), int(pyref.EQ - 1).to_bytes(32, "little")]
    y = 2
    while pyref.ed_decompress(int(y).to_bytes(32, "little")) is not None or pow((y * y - 1) * pow(pyref.ED_D * y * y + 1, -1, pyref.EQ) % pyref.EQ, (pyref.EQ - 1) // 2, pyref.EQ) == 1:
        y += 1
    bad.append(int(y).to_bytes(32, "little"))                      # non-residue
    y = 3
    while True:                                                     # on the curve but with a cofactor component
        w = (y * y - 1) * pow(pyref.ED_D * y * y + 1, -1, pyref.EQ) % pyref.EQ
        if pow(w, (pyref.EQ - 1) // 2, pyref.EQ) == 1 and pyref.ed_decompress(int(y).to_bytes(32, "little")) is None:
            break
        y += 1
    bad.append(int(y).to_bytes(32, "little"))
    data = np.frombuffer(b"".join(good + bad), dtype=np.uint8).copy()
    out, ok = oracle.ed_from_bytes(data)
    assert ok.tolist() == [1] * len(good) + [0] * len(bad)
    assert [pyref.ed_decompress(b) for b in good] == pts and all(pyref.ed_decompress(b) is None for b in bad)
    xy = oracle.ed_batch_to_affine(out[:16 * len(good)])
    got = []
    for i in range(len(good)):
        x, yv = limbs_to_ints(xy[8 * i:8 * i + 8])
        got.append((pyref.from_mont(4, x), pyref.from_mont(4, yv)))
    assert got == pts
