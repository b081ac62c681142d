#!/usr/bin/env python3
"""Writes a file in the SCHEMA of tools/ref_vectors/src/main.rs from this repo's Python model (tests/pyref.py): exact integers, hashlib,
the affine group laws, the dummy source's constants.  It is NOT a reference output -- it exists so that tests/test_ref_vectors.py (the
consumer of the real tests/golden/ref_vectors.json) is itself exercised in this image, where no Rust toolchain can produce the real
file: when someone runs the cargo recipe, the consumer is known to parse and check every field.

usage: python tools/ref_vectors/model_vectors.py out.json"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import pyref  # noqa: E402

FID = {"bn254": 0, "curve25519": 2}


def test_scalars(fid):
    p = pyref.P[fid]
    v = [0, 1, 2, p - 1, p - 2, (1 << 64) % p, ((1 << 128) + 12345) % p, ((1 << 192) - 1) % p, ((1 << 250) + 7) % p]
    x = 0xA11CE001 % p
    for _ in range(8):
        x = (x * x + 0x9E3779B97F4A7C15) % p
        v.append(x)
    return v


def mul_g(curve, k):
    return pyref.g1_mul(pyref.G, k) if curve == "bn254" else pyref.ed_mul(pyref.ED_B, k)


def compress(curve, pt):
    return pyref.g1_compress(pt) if curve == "bn254" else pyref.ed_compress(pt)


def neg(curve, pt):
    return pyref.g1_neg(pt) if curve == "bn254" else pyref.ed_neg(pt)


def add(curve, a, b):
    return pyref.g1_add(a, b) if curve == "bn254" else pyref.ed_add(a, b)


def serde_bytes(b):
    return json.dumps(list(b), separators=(",", ":"))


def build():
    doc = {"generator": "tools/ref_vectors/model_vectors.py: THIS REPO'S PYTHON MODEL in the recipe's schema, not a reference output",
           "scalars": [], "points": [], "wire": [], "commitments": []}
    for curve, fid in FID.items():
        p = pyref.P[fid]
        sc = test_scalars(fid)
        for s in sc:
            doc["scalars"].append({"curve": curve, "value_dec": str(s), "to_bytes_be_hex": pyref.to_bytes_be(fid, s).hex(),
                                   "serde_json": serde_bytes(s.to_bytes(32, "little"))})
            pt = mul_g(curve, s)
            doc["points"].append({"curve": curve, "scalar_dec": str(s), "to_bytes_hex": compress(curve, pt).hex(),
                                  "neg_to_bytes_hex": compress(curve, neg(curve, pt)).hex(),
                                  "double_to_bytes_hex": compress(curve, add(curve, pt, pt)).hex()})
        for rid, n in ((6, 0), (7, 1), (1234567, len(sc))):
            frame = pyref.wire_frame("ScalarBatch", rid, pyref.wire_scalar_records(fid, sc[:n]))
            doc["wire"].append({"curve": curve, "variant": "ScalarBatch", "result_id": rid, "values_dec": [str(s) for s in sc[:n]], "json": frame[8:].decode()})
        pts = [compress(curve, mul_g(curve, k)) for k in sc[:5]]
        frame = pyref.wire_frame("PointBatch", 99, pts)
        doc["wire"].append({"curve": curve, "variant": "PointBatch", "result_id": 99, "scalars_dec": [str(s) for s in sc[:5]],
                            "points_to_bytes_hex": [b.hex() for b in pts], "json": frame[8:].decode()})
        import hashlib
        for n in (0, 1, 4, len(sc)):
            blinder = sc[len(sc) - 1 - (n % 3)]
            h = hashlib.sha3_256(b"".join(pyref.to_bytes_be(fid, v) for v in sc[:n]) + pyref.to_bytes_be(fid, blinder)).digest()
            doc["commitments"].append({"curve": curve, "values_dec": [str(s) for s in sc[:n]], "blinder_dec": str(blinder), "sha3_256_hex": h.hex(),
                                       "commitment_dec": str(int.from_bytes(h, "big") % p)})
    # batch_mul under PartyIDBeaverSource (offline_prep.rs:103-170): share_scalar(v) -> P0 (v - 3, 0), P1 (3, v);
    # product with d = x - 2, e = y - 3 -> P0 (3d + e + 2 + de, 0), P1 (e + 4, 3d + 2e + 6 + de)
    p = pyref.P[0]
    xs = test_scalars(0)
    ys = xs[::-1]
    rows = [[], []]
    for x, y in zip(xs, ys):
        d, e = (x - 2) % p, (y - 3) % p
        rows[0].append({"x_share": [str((x - 3) % p), "0"], "y_share": [str((y - 3) % p), "0"], "product_share": [str((3 * d + e + 2 + d * e) % p), "0"]})
        rows[1].append({"x_share": ["3", str(x)], "y_share": ["3", str(y)], "product_share": [str((e + 4) % p), str((3 * d + 2 * e + 6 + d * e) % p)]})
    opened = [str(x * y % p) for x, y in zip(xs, ys)]
    doc["batch_mul"] = {"curve": "bn254", "source": "PartyIDBeaverSource", "x_dec": [str(v) for v in xs], "y_dec": [str(v) for v in ys],
                        "party0": {"shares": rows[0], "opened_dec": opened}, "party1": {"shares": rows[1], "opened_dec": opened}}
    return doc


if __name__ == "__main__":
    json.dump(build(), open(sys.argv[1], "w"), indent=1)
