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

FID = {"bn254": 0, "curve25519": 2, "bls12_381": 1}
POINT_CURVES = ("bn254", "curve25519")          # BLS12-381 is on the path as a scalar field only (config 5)


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
            if curve not in POINT_CURVES:
                continue
            pt = mul_g(curve, s)
            doc["points"].append({"curve": curve, "scalar_dec": str(s), "to_bytes_hex": compress(curve, pt).hex(),
                                  "neg_to_bytes_hex": compress(curve, neg(curve, pt)).hex(),
                                  "double_to_bytes_hex": compress(curve, add(curve, pt, pt)).hex()})
        for rid, n in ((6, 0), (7, 1), (1234567, len(sc))):
            frame = pyref.wire_frame("ScalarBatch", rid, pyref.wire_scalar_records(fid, sc[:n]))
            doc["wire"].append({"curve": curve, "variant": "ScalarBatch", "result_id": rid, "values_dec": [str(s) for s in sc[:n]], "json": frame[8:].decode()})
        if curve in POINT_CURVES:
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
    # ---- schema 2: the sections produced under the recipe's FixedSource (src/main.rs: the same closed formulas) -----------------------
    doc["schema"] = 2
    doc["ark_mpc_rev"] = "none: python model"
    doc["batch_mul_fixed"] = [batch_mul_fixed(c) for c in ("bn254", "curve25519", "bls12_381")]
    doc["open_authenticated"] = [open_authenticated(c) for c in ("bls12_381", "bn254")]
    doc["point_mul"] = [point_mul(c) for c in POINT_CURVES]
    return doc


class FixedSource:
    """tools/ref_vectors/src/main.rs FixedSource<C>, formula for formula"""
    K = (0x1234567, 0x89ABCDEF)

    def __init__(self, fid, party):
        self.fid, self.p, self.party, self.ctr = fid, pyref.P[fid], party, {}
        self.key = sum(self.K) % self.p

    def val(self, tag, i):
        b = ((tag << 32) + i + 1) % self.p
        return (b * b * b + 0x9E3779B97F4A7C15) % self.p

    def split(self, tag, i, v):
        r, m = self.val(tag + 16, i), self.val(tag + 32, i)
        return (r, m) if self.party == 0 else ((v - r) % self.p, (self.key * v - m) % self.p)

    def nxt(self, tag):
        i = self.ctr.get(tag, 0)
        self.ctr[tag] = i + 1
        return i

    def mask(self):                       # next_local_input_mask / next_counterparty_input_mask: one sequence
        i = self.nxt(3)
        v = self.val(3, i)
        return v, self.split(3, i, v)

    def triple(self):
        i = self.nxt(1)
        a, b = self.val(1, i), self.val(2, i)
        return self.split(1, i, a), self.split(2, i, b), self.split(8, i, a * b % self.p)


def share_scalars(src, vals):
    """fabric.rs:578-600 for one party: the sender broadcasts v - mask; both sides do mask_share.add_public(masked) (share.rs:74-77)"""
    p, out = src.p, []
    for v in vals:
        m, (s, mac) = src.mask()
        masked = (v - m) % p
        out.append(((s + masked) % p if src.party == 0 else s, (mac + src.K[src.party] * masked) % p))
    return out


def beaver(src, xs, ys, x_sh, y_sh, tri):
    """authenticated_scalar.rs:848-879 for one party, d and e from the plaintexts (d = x - a, e = y - b with a, b the triple's values)"""
    p, out = src.p, []
    for i, (x, y) in enumerate(zip(xs, ys)):
        (a_s, a_m), (b_s, b_m), (c_s, c_m) = tri[i]
        d, e = (x - src.val(1, i)) % p, (y - src.val(2, i)) % p
        de = d * e % p
        out.append(((d * b_s + e * a_s + c_s + (de if src.party == 0 else 0)) % p, (d * b_m + e * a_m + c_m + src.K[src.party] * de) % p))
    return out


def sh(pair):
    return [str(pair[0]), str(pair[1])]


def batch_mul_fixed(curve):
    fid = FID[curve]
    p = pyref.P[fid]
    xs = test_scalars(fid)
    ys = xs[::-1]
    parties = []
    for party in (0, 1):
        src = FixedSource(fid, party)
        x_sh, y_sh = share_scalars(src, xs), share_scalars(src, ys)
        tri = [src.triple() for _ in xs]
        prod = beaver(src, xs, ys, x_sh, y_sh, tri)
        rows = [{"x_share": sh(x_sh[i]), "y_share": sh(y_sh[i]), "a_share": sh(tri[i][0]), "b_share": sh(tri[i][1]), "c_share": sh(tri[i][2]),
                 "product_share": sh(prod[i])} for i in range(len(xs))]
        parties.append({"key_share_dec": str(src.K[party]), "shares": rows, "opened_dec": [str(x * y % p) for x, y in zip(xs, ys)]})
    return {"curve": curve, "source": "FixedSource", "x_dec": [str(v) for v in xs], "y_dec": [str(v) for v in ys], "party0": parties[0], "party1": parties[1]}


def open_authenticated(curve):
    import hashlib
    fid = FID[curve]
    p = pyref.P[fid]
    vs = test_scalars(fid)
    parties = []
    for party in (0, 1):
        src = FixedSource(fid, party)
        local = share_scalars(src, vs)
        chk = [(src.K[party] * v - mac) % p for v, (_, mac) in zip(vs, local)]
        blinder = 0xB11D0000 + party
        h = hashlib.sha3_256(b"".join(pyref.to_bytes_be(fid, c) for c in chk) + pyref.to_bytes_be(fid, blinder)).digest()
        parties.append({"key_share_dec": str(src.K[party]), "shares": [sh(s) for s in local], "opened_dec": [str(v) for v in vs],
                        "mac_check_shares_dec": [str(c) for c in chk], "blinder_dec": str(blinder), "mac_check_commitment_dec": str(int.from_bytes(h, "big") % p)})
    return {"curve": curve, "source": "FixedSource", "values_dec": [str(v) for v in vs], "party0": parties[0], "party1": parties[1]}


def point_mul(curve):
    fid = FID[curve]
    p = pyref.P[fid]
    sc = test_scalars(fid)
    ss, ks, xs = sc[1:9], sc[::-1][:8], sc[5:13]
    ident = mul_g(curve, 0)
    G = lambda k: mul_g(curve, k % p)
    mulp = lambda pt, k: (pyref.g1_mul(pt, k % p) if curve == "bn254" else pyref.ed_mul(pt, k % p))
    sub = lambda a_, b_: add(curve, a_, neg(curve, b_))
    hexs = lambda ps_: [compress(curve, ps_[0]).hex(), compress(curve, ps_[1]).hex()]
    parties = []
    for party in (0, 1):
        src = FixedSource(fid, party)
        k_p = src.K[party]
        # fabric.rs:622-649: the sender broadcasts P - mask * G; both sides compute mask_share * G + masked (curve/share.rs:57-60)
        pt_sh = []
        for s_ in ss:
            m, (ms, mm) = src.mask()
            masked = sub(G(s_), G(m))
            pt_sh.append((add(curve, G(ms), masked) if party == 0 else G(ms), add(curve, G(mm), mulp(masked, k_p))))
        x_sh = share_scalars(src, xs)
        tri = [src.triple() for _ in ss]
        rows, opened_pub, opened_bv = [], [], []
        for i in range(len(ss)):
            (a_s, a_m), (b_s, b_m), (c_s, c_m) = tri[i]
            mul_pub = (mulp(pt_sh[i][0], ks[i]), mulp(pt_sh[i][1], ks[i]))                                  # curve/share.rs:108-114
            # authenticated_curve.rs:682-714 with d = x - a, eG = (s - b) G opened: deG + d[bG] + [a]eG + [c]G
            d = (xs[i] - src.val(1, i)) % p
            eG = G(ss[i] - src.val(2, i))
            deG = mulp(eG, d)
            share = add(curve, add(curve, G(d * b_s), mulp(eG, a_s)), G(c_s))
            if party == 0:
                share = add(curve, share, deG)
            mac = add(curve, add(curve, add(curve, G(d * b_m), mulp(eG, a_m)), G(c_m)), mulp(deG, k_p))
            rows.append({"point_share": hexs(pt_sh[i]), "x_share": sh(x_sh[i]), "a_share": sh(tri[i][0]), "b_share": sh(tri[i][1]), "c_share": sh(tri[i][2]),
                         "mul_public_share": hexs(mul_pub), "beaver_mul_share": hexs((share, mac))})
            opened_pub.append(compress(curve, G(ss[i] * ks[i])).hex())
            opened_bv.append(compress(curve, G(ss[i] * xs[i])).hex())
        parties.append({"key_share_dec": str(k_p), "rows": rows, "opened_mul_public_hex": opened_pub, "opened_beaver_mul_hex": opened_bv})
    del ident
    return {"curve": curve, "source": "FixedSource", "point_scalars_dec": [str(v) for v in ss], "public_scalars_dec": [str(v) for v in ks],
            "shared_scalars_dec": [str(v) for v in xs], "party0": parties[0], "party1": parties[1]}


if __name__ == "__main__":
    json.dump(build(), open(sys.argv[1], "w"), indent=1)
