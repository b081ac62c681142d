#!/usr/bin/env python3
"""Generates tests/golden/*.json -- small known-answer vectors for the hot path.

The reference (renegade-fi/ark-mpc) holds no golden vectors and cannot be run in the build image
(Rust + un-vendored arkworks), so these come from exact integer arithmetic in Python (the unique
canonical residues any correct field implementation must produce), hashlib.sha3_256, the affine
BN254 G1 group law, and the reference's constant-valued PartyIDBeaverSource
(online-phase/src/offline_prep.rs:103-170).  All field values are stored as canonical hex integers;
the tests convert to arkworks' Montgomery limb layout.

Run from the repo root:  python tests/golden/gen_golden.py
"""
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import pyref  # noqa: E402

hx = lambda v: hex(v)


def field_vectors():
    out = {}
    for fid, p in pyref.P.items():
        rng = random.Random(0xA11CE000 + fid)
        edge = [0, 1, 2, p - 1, p - 2, (1 << 255) % p, (1 << 192) - 1, 1 << 64, (p + 1) // 2]
        a = edge + [rng.randrange(p) for _ in range(23)]
        b = list(reversed(edge)) + [rng.randrange(p) for _ in range(23)]
        out[str(fid)] = {
            "modulus": hx(p), "a": [hx(v) for v in a], "b": [hx(v) for v in b],
            "add": [hx((x + y) % p) for x, y in zip(a, b)], "sub": [hx((x - y) % p) for x, y in zip(a, b)],
            "mul": [hx((x * y) % p) for x, y in zip(a, b)], "neg": [hx((-x) % p) for x in a],
            "mont_a": [hx(pyref.to_mont(fid, v)) for v in a],
            "bytes_be_a": [pyref.to_bytes_be(fid, v).hex() for v in a],
        }
    return out


def dummy_source_vectors():
    """Known answers derivable from PartyIDBeaverSource: MAC key shares
    k0 = 0, k1 = 1; triple a = 2, b = 3, c = 6 with P0 (1,0),(3,0),(2,0) and P1 (1,2),(0,3),(4,6) as (share, mac);
    input masks: cleartext 3, party p's mask share (3p, 3p)."""
    out = {}
    for fid in (0, 1, 2):
        p = pyref.P[fid]
        rng = random.Random(0xD00D + fid)
        xs = [0, 1, 5, p - 1] + [rng.randrange(p) for _ in range(12)]
        ys = [7, 0, p - 1, p - 1] + [rng.randrange(p) for _ in range(12)]
        cases = []
        for x, y in zip(xs, ys):
            # share_scalar(v, sender) for either sender: P0 holds (v - 3, 0), P1 holds (3, v)  (fabric.rs:560-574)
            sx = [((x - 3) % p, 0), (3, x % p)]
            sy = [((y - 3) % p, 0), (3, y % p)]
            d, e = (x - 2) % p, (y - 3) % p
            de = d * e % p
            out0 = ((3 * d + e + 2 + de) % p, 0)                    # P0: d*3 + e*1 + 2 + de ; mac 0
            out1 = ((e + 4) % p, (3 * d + 2 * e + 6 + de) % p)      # P1: d*0 + e*1 + 4     ; mac d*3 + e*2 + 6 + 1*de
            assert (out0[0] + out1[0]) % p == x * y % p and (out0[1] + out1[1]) % p == x * y % p
            cases.append({
                "x": hx(x), "y": hx(y),
                "x_shares": [[hx(s), hx(m)] for s, m in sx], "y_shares": [[hx(s), hx(m)] for s, m in sy],
                "d_shares": [hx((sx[0][0] - 1) % p), hx((sx[1][0] - 1) % p)],
                "e_shares": [hx((sy[0][0] - 3) % p), hx((sy[1][0] - 0) % p)],
                "d": hx(d), "e": hx(e),
                "out_shares": [[hx(out0[0]), hx(out0[1])], [hx(out1[0]), hx(out1[1])]],
                "product": hx(x * y % p),
                # MAC-check shares on opening the product v with P1 share-mac m: P0 0*v - 0, P1 1*v - m
                "mac_check_shares": [hx(0), hx((x * y - out1[1]) % p)],
            })
        out[str(fid)] = {"mac_key_shares": [hx(0), hx(1)],
                         "triple_shares": {"p0": {"a": [hx(1), hx(0)], "b": [hx(3), hx(0)], "c": [hx(2), hx(0)]},
                                           "p1": {"a": [hx(1), hx(2)], "b": [hx(0), hx(3)], "c": [hx(4), hx(6)]}},
                         "cases": cases}
    return out


def commitment_vectors():
    out = {}
    for fid in (0, 1, 2):
        p = pyref.P[fid]
        rng = random.Random(0xC0FFEE + fid)
        cases = []
        for n in (1, 2, 4, 5, 17, 64):
            vals = [rng.randrange(p) for _ in range(n)]
            if n >= 4:
                vals[0], vals[1] = 0, p - 1
            blinder = rng.randrange(p)
            cases.append({"values": [hx(v) for v in vals], "blinder": hx(blinder), "commitment": hx(pyref.commit(fid, vals, blinder))})
        out[str(fid)] = cases
    return out


def curve_vectors():
    rng = random.Random(0xEC)
    r = pyref.RORD
    ks = [0, 1, 2, 3, r - 1, r - 2, (1 << 253) % r, 0xFFFFFFFF, 1 << 128] + [rng.randrange(r) for _ in range(7)]
    base = [1, 1, 5, 7, 11, 1, 2, 3, 1] + [rng.randrange(r) for _ in range(7)]
    pts = [pyref.g1_mul(pyref.G, b) for b in base]
    enc = lambda P: None if P is None else [hx(P[0]), hx(P[1])]
    return {
        "generator": enc(pyref.G),
        "two_g_eip196": enc(pyref.g1_add(pyref.G, pyref.G)),
        "base_scalars": [hx(b) for b in base], "points": [enc(P) for P in pts], "scalars": [hx(k) for k in ks],
        "scalar_mul": [enc(pyref.g1_mul(P, k)) for P, k in zip(pts, ks)],
        "add_reversed": [enc(pyref.g1_add(P, Q)) for P, Q in zip(pts, reversed(pts))],
        "double": [enc(pyref.g1_add(P, P)) for P in pts],
        "add_neg": [enc(pyref.g1_add(P, pyref.g1_neg(P))) for P in pts],
        "compressed": [pyref.g1_compress(P).hex() for P in pts],
        "compressed_identity": pyref.g1_compress(None).hex(),
    }


def msm_vectors():
    """CurvePoint::msm / msm_authenticated (curve.rs:549-560, 618-642) and CurvePoint::from_bytes (:110-114)."""
    rng = random.Random(0x3535)
    r, q = pyref.RORD, pyref.Q
    n = 24
    base = [rng.randrange(r) for _ in range(n)]
    pts = [pyref.g1_mul(pyref.G, b) for b in base]
    pts[3] = None                                   # identity among the bases
    pts[9] = pts[8]                                 # repeated base
    scalars = [0, 1, r - 1, 2, (1 << 253) % r] + [rng.randrange(r) for _ in range(n - 5)]
    scalars[9] = scalars[8]                         # ... with the same scalar: doubling inside a bucket
    macs = [rng.randrange(r) for _ in range(n)]
    enc = lambda P: None if P is None else [hx(P[0]), hx(P[1])]
    def msm(ss):
        acc = None
        for k, P in zip(ss, pts):
            acc = pyref.g1_add(acc, pyref.g1_mul(P, k) if P else None)
        return acc
    # encodings that are NOT points: x >= q, both flag bits, x^3 + 3 a non-residue
    x = 0
    while pow((x ** 3 + 3) % q, (q - 1) // 2, q) == 1 or (x ** 3 + 3) % q == 0:
        x += 1
    bad = [int(q).to_bytes(32, "little").hex(), (bytes(31) + b"\xc0").hex(), int(x).to_bytes(32, "little").hex()]
    return {"points": [enc(P) for P in pts], "scalars": [hx(k) for k in scalars], "macs": [hx(k) for k in macs],
            "msm": enc(msm(scalars)), "msm_macs": enc(msm(macs)),
            "compressed": [pyref.g1_compress(P).hex() for P in pts], "invalid_encodings": bad}


def wire_vectors():
    """QuicTwoPartyNet frames (network/quic.rs:303-306): u64 LE length + serde_json text, see pyref.wire_frame."""
    out = {"scalar_batches": [], "point_batches": []}
    for fid in (0, 1, 2):
        p = pyref.P[fid]
        rng = random.Random(0x77 + fid)
        for n, rid in ((0, 0), (1, 7), (5, 4294967296), (40, 18446744073709551615)):
            vals = ([0, 1, p - 1, 255, 256] + [rng.randrange(p) for _ in range(n)])[:n]
            out["scalar_batches"].append({"field": fid, "result_id": rid, "values": [hx(v) for v in vals],
                                          "frame": pyref.wire_frame("ScalarBatch", rid, pyref.wire_scalar_records(fid, vals)).hex()})
    rng = random.Random(0x78)
    pts = [pyref.g1_mul(pyref.G, rng.randrange(pyref.RORD)) for _ in range(6)] + [None]
    out["point_batches"].append({"result_id": 12, "points": [None if P is None else [hx(P[0]), hx(P[1])] for P in pts],
                                 "frame": pyref.wire_frame("PointBatch", 12, [pyref.g1_compress(P) for P in pts]).hex()})
    good = pyref.wire_frame("ScalarBatch", 3, pyref.wire_scalar_records(0, [5, 255 * 256, 77]))[8:]
    import struct
    fr = lambda b: (struct.pack("<Q", len(b)) + b).hex()
    out["malformed_scalar_batches_field0"] = {
        "length prefix": (struct.pack("<Q", len(good) + 1) + good).hex(),
        "whitespace inside a number": fr(good.replace(b",255,", b",2 55,", 1)),
        "byte > 255": fr(good.replace(b"[[5,", b"[[256,", 1)),
        "leading zero": fr(good.replace(b",255,", b",0255,", 1)),
        "31 numbers": fr(good.replace(b",255,", b",", 1)),
        "33 numbers": fr(good.replace(b",255,", b",255,255,", 1)),
        "other variant": fr(good.replace(b"ScalarBatch", b"ScalarShare", 1)),
        "scalar >= modulus": pyref.wire_frame("ScalarBatch", 3, [int(pyref.P[0]).to_bytes(32, "little")]).hex(),
    }
    # serde_json::from_slice skips whitespace between tokens: accepted, same values
    out["whitespace_scalar_batches_field0"] = {"values": [hx(v) for v in (5, 255 * 256, 77)], "result_id": 3,
                                               "frames": [fr(good.replace(b"[[", b"[ [", 1)), fr(good.replace(b",", b" ,\n\t").replace(b":", b" : ") + b"\r\n")]}
    return out


if __name__ == "__main__":
    for name, fn in (("field_vectors", field_vectors), ("dummy_source_vectors", dummy_source_vectors),
                     ("commitment_vectors", commitment_vectors), ("curve_vectors", curve_vectors), ("msm_vectors", msm_vectors),
                     ("wire_vectors", wire_vectors)):
        with open(os.path.join(HERE, name + ".json"), "w") as f:
            json.dump(fn(), f, indent=1)
        print("wrote", name + ".json")
