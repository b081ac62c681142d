"""Consumer of tests/golden/ref_vectors.json -- vectors emitted by the REFERENCE itself through tools/ref_vectors (cargo; see its README).
The build image has no Rust toolchain, so the file is absent here and these tests SKIP; once someone runs the recipe, the same checks run
against the oracle (CPU) and, with -m gpu, against the HIP engine through the C ABI, and the repo's parity is pinned by the reference.

So that the consumer itself is not dead code until then, test_consumer_on_model_file runs it on a file in the recipe's schema written by
this repo's Python model (tools/ref_vectors/model_vectors.py) -- that proves the consumer, not parity."""
import json
import os
import struct
import subprocess
import sys

import numpy as np
import pytest

import pyref
from helpers import EngineAdapter, ints_to_limbs, limbs_to_ints, mont_array, from_mont_array, interleave_shares

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REAL = os.path.join(ROOT, "tests", "golden", "ref_vectors.json")
FID = {"bn254": 0, "curve25519": 2, "bls12_381": 1}        # schema 2 adds BLS12-381 Fr (BASELINE config 5's field)
POINT_CURVES = ("bn254", "curve25519")


def load_real():
    if not os.path.exists(REAL):
        pytest.skip("tests/golden/ref_vectors.json absent: run tools/ref_vectors (needs cargo) to pin parity with reference-generated vectors")
    return json.load(open(REAL))


def check_doc(be, doc, points_through_backend=True):
    """`be` = the oracle binding or EngineAdapter(HIP): same method shapes.  Returns the number of values compared."""
    cnt = 0
    # --- scalars: Scalar::to_bytes_be (K6) and the 32 LE bytes of serialize_uncompressed
    for fid_name in FID:
        fid = FID[fid_name]
        rows = [r for r in doc["scalars"] if r["curve"] == fid_name]
        if not rows:
            continue                                                         # (a schema-1 file has no BLS12-381 rows)
        vals = [int(r["value_dec"]) for r in rows]
        got = be.to_bytes_be(fid, mont_array(fid, vals)).tobytes()
        assert got == b"".join(bytes.fromhex(r["to_bytes_be_hex"]) for r in rows)
        canon = be.to_canonical(fid, mont_array(fid, vals))                   # canonical LE limbs = the bytes serde carries
        for r, row in zip(rows, np.asarray(canon, dtype=np.uint64).reshape(-1, 4)):
            assert list(row.tobytes()) == json.loads(r["serde_json"])
        cnt += 2 * len(rows)
    # --- points: k * G compressed, the negation and the doubling (flag bits: SW y > -y / infinity, TE x > -x)
    for fid_name in POINT_CURVES:
        rows = [r for r in doc["points"] if r["curve"] == fid_name]
        ks = [int(r["scalar_dec"]) for r in rows]
        n = len(ks)
        if fid_name == "bn254":
            G = np.tile(np.array(pyref.g1_jacobian_mont(pyref.G), dtype=np.uint64), n)
            P = be.g1_batch_scalar_mul(G, mont_array(0, ks))
            assert be.g1_to_bytes(P).tobytes() == b"".join(bytes.fromhex(r["to_bytes_hex"]) for r in rows)
            assert be.g1_to_bytes(be.g1_neg(P)).tobytes() == b"".join(bytes.fromhex(r["neg_to_bytes_hex"]) for r in rows)
            assert be.g1_to_bytes(be.g1_batch_add(P, P)).tobytes() == b"".join(bytes.fromhex(r["double_to_bytes_hex"]) for r in rows)
        else:
            B = np.tile(np.array(pyref.ed_extended_mont(pyref.ED_B), dtype=np.uint64), n)
            P = be.ed_batch_scalar_mul(B, mont_array(2, ks))
            assert be.ed_to_bytes(P).tobytes() == b"".join(bytes.fromhex(r["to_bytes_hex"]) for r in rows)
            assert be.ed_to_bytes(be.ed_batch_neg(P)).tobytes() == b"".join(bytes.fromhex(r["neg_to_bytes_hex"]) for r in rows)
            assert be.ed_to_bytes(be.ed_batch_add(P, P)).tobytes() == b"".join(bytes.fromhex(r["double_to_bytes_hex"]) for r in rows)
        cnt += 3 * n
    # --- commitments: H1
    for r in doc["commitments"]:
        fid = FID[r["curve"]]
        vals = [int(v) for v in r["values_dec"]]
        got = be.commit_scalars(fid, mont_array(fid, vals), mont_array(fid, [int(r["blinder_dec"])]))
        assert from_mont_array(fid, got) == [int(r["commitment_dec"])]
        cnt += 1
    # --- batch_mul under PartyIDBeaverSource: local shares of inputs and products, and the opening
    bm = doc["batch_mul"]
    fid = FID[bm["curve"]]
    p = pyref.P[fid]
    n = len(bm["x_dec"])
    rec = lambda rows, key: interleave_shares(mont_array(fid, [int(r[key][0]) for r in rows]), mont_array(fid, [int(r[key][1]) for r in rows]))
    keys = [mont_array(fid, [0]), mont_array(fid, [1])]                      # offline_prep.rs:108-110: mac key share = party id
    tri = {0: {"a": (1, 0), "b": (3, 0), "c": (2, 0)}, 1: {"a": (1, 2), "b": (0, 3), "c": (4, 6)}}   # offline_prep.rs:137-158
    sh = {}
    for pid, party in ((0, bm["party0"]), (1, bm["party1"])):
        rows = party["shares"]
        sh[pid] = {"x": rec(rows, "x_share"), "y": rec(rows, "y_share"), "want": rec(rows, "product_share")}
        for nm, (s, m) in tri[pid].items():
            sh[pid][nm] = interleave_shares(mont_array(fid, [s] * n), mont_array(fid, [m] * n))
    de = [be.beaver_mask(fid, sh[q]["x"], sh[q]["y"], sh[q]["a"], sh[q]["b"]) for q in (0, 1)]
    opened_de = be.open_combine(fid, de[0], de[1])
    for q in (0, 1):
        got = be.beaver_finish(fid, q, keys[q], opened_de[:4 * n].copy(), opened_de[4 * n:].copy(), sh[q]["a"], sh[q]["b"], sh[q]["c"])
        assert np.array_equal(got, sh[q]["want"]), "party %d product shares differ from the reference's" % q
    ext = lambda a: np.ascontiguousarray(a.reshape(-1, 8)[:, :4]).reshape(-1)
    opened = be.open_combine(fid, ext(sh[0]["want"]), ext(sh[1]["want"]))
    assert [str(v) for v in from_mont_array(fid, opened)] == bm["party0"]["opened_dec"] == bm["party1"]["opened_dec"]
    chk = [be.mac_check_shares(fid, keys[q], opened, sh[q]["want"]) for q in (0, 1)]
    assert be.mac_verify(fid, chk[0], chk[1])
    cnt += 3 * n
    if doc.get("schema", 1) >= 2:
        cnt += check_schema2(be, doc)
    return cnt


def _records(fid, rows, key):
    return interleave_shares(mont_array(fid, [int(r[key][0]) for r in rows]), mont_array(fid, [int(r[key][1]) for r in rows]))


def _share_half(rec):
    return np.ascontiguousarray(rec.reshape(-1, 8)[:, :4]).reshape(-1)


def check_schema2(be, doc):
    """the sections produced under the recipe's non-degenerate FixedSource: batch_mul on all three scalar fields of the BASELINE configs, config 5's
    intermediates (opened values, MAC-check shares, their commitment), config 4's point gates.  Everything the backend needs is in the file."""
    cnt = 0
    # --- batch_mul: each party's local product shares from its local input / triple shares and the two opened masks
    for bm in doc["batch_mul_fixed"]:
        fid = FID[bm["curve"]]
        n = len(bm["x_dec"])
        P = [bm["party0"], bm["party1"]]
        keys = [mont_array(fid, [int(q["key_share_dec"])]) for q in P]
        R = [{nm: _records(fid, q["shares"], nm + "_share") for nm in ("x", "y", "a", "b", "c", "product")} for q in P]
        de = [be.beaver_mask(fid, R[q]["x"], R[q]["y"], R[q]["a"], R[q]["b"]) for q in (0, 1)]
        opened_de = be.open_combine(fid, de[0], de[1])
        for q in (0, 1):
            got = be.beaver_finish(fid, q, keys[q], opened_de[:4 * n].copy(), opened_de[4 * n:].copy(), R[q]["a"], R[q]["b"], R[q]["c"])
            assert np.array_equal(got, R[q]["product"]), "%s: party %d product shares differ from the reference's" % (bm["curve"], q)
        opened = be.open_combine(fid, _share_half(R[0]["product"]), _share_half(R[1]["product"]))
        assert [str(v) for v in from_mont_array(fid, opened)] == P[0]["opened_dec"] == P[1]["opened_dec"]
        chk = [be.mac_check_shares(fid, keys[q], opened, R[q]["product"]) for q in (0, 1)]
        assert be.mac_verify(fid, chk[0], chk[1])
        cnt += 3 * n
    # --- open_authenticated_batch: opened values, this party's MAC-check shares and their commitment
    for oa in doc["open_authenticated"]:
        fid = FID[oa["curve"]]
        P = [oa["party0"], oa["party1"]]
        n = len(oa["values_dec"])
        keys = [mont_array(fid, [int(q["key_share_dec"])]) for q in P]
        R = [interleave_shares(mont_array(fid, [int(s_[0]) for s_ in q["shares"]]), mont_array(fid, [int(s_[1]) for s_ in q["shares"]])) for q in P]
        opened = be.open_combine(fid, _share_half(R[0]), _share_half(R[1]))
        assert [str(v) for v in from_mont_array(fid, opened)] == P[0]["opened_dec"] == P[1]["opened_dec"] == oa["values_dec"]
        chk = [be.mac_check_shares(fid, keys[q], opened, R[q]) for q in (0, 1)]
        for q in (0, 1):
            assert [str(v) for v in from_mont_array(fid, chk[q])] == P[q]["mac_check_shares_dec"]
            com = be.commit_scalars(fid, chk[q], mont_array(fid, [int(P[q]["blinder_dec"])]))
            assert from_mont_array(fid, com) == [int(P[q]["mac_check_commitment_dec"])]
        assert be.mac_verify(fid, chk[0], chk[1])
        cnt += 3 * n
    # --- config 4: PointShare x public Scalar on both curves; the full Beaver point gate on BN254 G1 in the reference's literal op sequence
    for pm in doc["point_mul"]:
        curve = pm["curve"]
        P = [pm["party0"], pm["party1"]]
        n = len(pm["point_scalars_dec"])
        fid = FID[curve]
        ks = mont_array(fid, [int(v) for v in pm["public_scalars_dec"]])
        frm = (lambda b_: be.g1_from_bytes(b_)) if curve == "bn254" else (lambda b_: be.ed_from_bytes(b_))
        tob = be.g1_to_bytes if curve == "bn254" else be.ed_to_bytes
        smul = be.g1_batch_scalar_mul if curve == "bn254" else be.ed_batch_scalar_mul
        addp = be.g1_batch_add if curve == "bn254" else be.ed_batch_add
        hexcol = lambda rows, key, j: np.frombuffer(b"".join(bytes.fromhex(r[key][j]) for r in rows), dtype=np.uint8).copy()
        halves = []
        for q in (0, 1):
            rows = P[q]["rows"]
            for j in (0, 1):                                                  # the share point and the MAC point: (k * share, k * mac), curve/share.rs:108-114
                pts, ok = frm(hexcol(rows, "point_share", j))
                assert ok.all()
                assert tob(smul(pts, ks)).tobytes() == hexcol(rows, "mul_public_share", j).tobytes(), (curve, q, j)
                if j == 0:
                    halves.append(smul(pts, ks))
        want_open = b"".join(bytes.fromhex(h) for h in P[0]["opened_mul_public_hex"])
        assert tob(addp(halves[0], halves[1])).tobytes() == want_open and P[0]["opened_mul_public_hex"] == P[1]["opened_mul_public_hex"]
        cnt += 5 * n
        if curve != "bn254":
            continue
        W = 12
        keys = [mont_array(0, [int(q["key_share_dec"])]) for q in P]
        R = [{nm: _records(0, q["rows"], nm + "_share") for nm in ("x", "a", "b", "c")} for q in P]

        def pshare(q, key):
            sp, ok1 = be.g1_from_bytes(hexcol(P[q]["rows"], key, 0)); mp, ok2 = be.g1_from_bytes(hexcol(P[q]["rows"], key, 1))
            assert ok1.all() and ok2.all()
            return np.ascontiguousarray(np.concatenate([sp.reshape(n, W), mp.reshape(n, W)], axis=1).reshape(-1))

        Y = [pshare(q, "point_share") for q in (0, 1)]
        bG = [be.scalarshare_mul_generator(R[q]["b"]) for q in (0, 1)]                              # beaver_b_gen               :698
        masked = [be.pointshare_add(Y[q], bG[q], sub=True) for q in (0, 1)]                          # masked_lhs = b - beaver_b_gen :701
        first = lambda ps_: np.ascontiguousarray(ps_.reshape(n, 2 * W)[:, :W]).reshape(-1)
        eG = be.g1_batch_add(first(masked[0]), first(masked[1]))                                     # open_batch(masked_lhs)     :703
        dsh = [be.scalar_sub(0, _share_half(R[q]["x"]), _share_half(R[q]["a"])) for q in (0, 1)]   # masked_rhs = a - beaver_a  :700
        d = be.open_combine(0, dsh[0], dsh[1])                                                       # open_batch(masked_rhs)     :704
        deG = be.g1_batch_scalar_mul(eG, d)                                                          # :707
        for q in (0, 1):
            dbG = be.pointshare_mul_public(bG[q], d)                                                 # :708
            aeG = be.scalarshare_mul_point(R[q]["a"], eG)                                            # :709
            cG = be.scalarshare_mul_generator(R[q]["c"])                                             # :710
            res = be.pointshare_add(be.pointshare_add_public(q, keys[q], dbG, deG), be.pointshare_add(aeG, cG))   # :712-714
            for j in (0, 1):
                half = np.ascontiguousarray(res.reshape(n, 2 * W)[:, j * W:(j + 1) * W]).reshape(-1)
                assert be.g1_to_bytes(half).tobytes() == hexcol(P[q]["rows"], "beaver_mul_share", j).tobytes(), ("beaver point mul", q, j)
        cnt += 2 * n
    return cnt


def check_wire_model(doc):
    """the frame text serde_json::to_vec produced == the Python wire model (which the GPU codec is tested against, tests/test_wire_format.py)"""
    cnt = 0
    for r in doc["wire"]:
        fid = FID[r["curve"]]
        if r["variant"] == "ScalarBatch":
            want = pyref.wire_frame("ScalarBatch", r["result_id"], pyref.wire_scalar_records(fid, [int(v) for v in r["values_dec"]]))
        else:
            want = pyref.wire_frame("PointBatch", r["result_id"], [bytes.fromhex(h) for h in r["points_to_bytes_hex"]])
        assert want[8:] == r["json"].encode() and struct.unpack("<Q", want[:8])[0] == len(r["json"])
        cnt += 1
    return cnt


def check_wire_hip(pkg, doc):
    """the HIP encoder writes the reference's text byte for byte, and the decoder reads it back"""
    hip = EngineAdapter(pkg)
    cnt = 0
    for r in doc["wire"]:
        if r["variant"] != "ScalarBatch":
            continue
        fid = FID[r["curve"]]
        vals = [int(v) for v in r["values_dec"]]
        n = len(vals)
        eng = hip.eng(fid)
        cap = eng.wire_frame_bound(n)
        buf = np.zeros(cap, dtype=np.uint8)
        ln = eng.wire_encode_scalar_batch(r["result_id"], n, mont_array(fid, vals) if n else np.zeros(4, dtype=np.uint64), buf, cap)
        assert buf[8:ln].tobytes() == r["json"].encode()
        out = np.zeros(4 * max(n, 1), dtype=np.uint64)
        got_n, rid = eng.wire_decode_scalar_batch(buf[:ln].copy(), ln, max(n, 1), out)
        assert (got_n, rid) == (n, r["result_id"]) and np.array_equal(out[:4 * n], mont_array(fid, vals))
        cnt += 1
    return cnt


def test_reference_vectors_on_oracle(oracle):
    doc = load_real()
    assert "PYTHON MODEL" not in doc.get("generator", ""), "tests/golden/ref_vectors.json must come from the cargo recipe, not from model_vectors.py"
    assert check_doc(oracle, doc) > 50 and check_wire_model(doc) >= 8


@pytest.mark.gpu
def test_reference_vectors_on_hip_engine(pkg):
    doc = load_real()
    assert check_doc(EngineAdapter(pkg), doc) > 50 and check_wire_hip(pkg, doc) >= 6


def _model_doc(tmp_path):
    out = tmp_path / "model_vectors.json"
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "ref_vectors", "model_vectors.py"), str(out)])
    return json.load(open(out))


def test_consumer_on_model_file(oracle, tmp_path):
    """The consumer parses and checks every field of the recipe's schema (on a model-generated file: proves the consumer, not parity)."""
    doc = _model_doc(tmp_path)
    assert check_doc(oracle, doc) > 50 and check_wire_model(doc) >= 8


@pytest.mark.gpu
def test_consumer_on_model_file_hip(pkg, tmp_path):
    doc = _model_doc(tmp_path)
    assert check_doc(EngineAdapter(pkg), doc) > 250 and check_wire_hip(pkg, doc) >= 9


def test_recipe_files_are_present():
    for f in ("Cargo.toml", "src/main.rs", "README.md", "rust-toolchain"):
        assert os.path.exists(os.path.join(ROOT, "tools", "ref_vectors", f))
    src = open(os.path.join(ROOT, "tools", "ref_vectors", "src", "main.rs")).read()
    for needle in ("to_bytes_be", "to_bytes()", "NetworkOutbound", "Sha3_256", "batch_mul", "execute_mock_mpc", "ark_bls12_381", "FixedSource", "batch_mul_public",
                   "open_authenticated_batch", "PreprocessingPhase<C> for FixedSource", "ARK_MPC_REV_RESOLVED"):
        assert needle in src
