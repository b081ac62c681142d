"""Wire format of the batches on the party-to-party link: the Python model used as the checker (pyref.wire_frame)
against a hand-written known answer, and -- on the GPU -- the HIP encoder / decoder against that model.
Reference: network.rs:33-60 (NetworkOutbound / NetworkPayload), network/quic.rs:303-306 (u64 LE length + serde_json),
scalar.rs:186-201 (Scalar <-> 32 canonical LE bytes, validated on read), curve.rs:50-63 (CurvePoint <-> compressed bytes)."""
import struct

import numpy as np
import pytest

import pyref
from helpers import mont_array, rand_values, EngineAdapter


def test_model_known_answer():
    """One-scalar ScalarBatch, value 1 + 255*256 + 2^248*10: the literal serde_json text."""
    v = 1 + 255 * 256 + (10 << 248)
    want = b'{"result_id":7,"payload":{"ScalarBatch":[[1,255' + b",0" * 29 + b",10]]}}"
    got = pyref.wire_frame("ScalarBatch", 7, pyref.wire_scalar_records(0, [v]))
    assert got == struct.pack("<Q", len(want)) + want
    assert pyref.wire_frame("PointBatch", 0, []) == struct.pack("<Q", 43) + b'{"result_id":0,"payload":{"PointBatch":[]}}'


@pytest.fixture(scope="module")
def hip(pkg):
    return EngineAdapter(pkg)


def encode(eng, rid, values_mont, n):
    cap = eng.wire_frame_bound(n)
    buf = np.zeros(cap, dtype=np.uint8)
    ln = eng.wire_encode_scalar_batch(rid, n, values_mont if n else np.zeros(4, dtype=np.uint64), buf, cap)
    return buf[:ln].tobytes()


@pytest.mark.gpu
@pytest.mark.parametrize("fid", [0, 1, 2])
@pytest.mark.parametrize("n", [0, 1, 2, 255, 256, 257, 3000])
def test_encode_scalar_batch_matches_model(hip, fid, n):
    p = pyref.P[fid]
    vals = ([0, 1, p - 1, 255, 256, 10 ** 20, (1 << 248) - 1] + rand_values(fid, max(n, 7), 500 + n))[:n]
    rid = [0, 9, 10, 12345678901234567890][n % 4]
    got = encode(hip.eng(fid), rid, mont_array(fid, vals), n)
    assert got == pyref.wire_frame("ScalarBatch", rid, pyref.wire_scalar_records(fid, vals))


@pytest.mark.gpu
@pytest.mark.parametrize("n", [0, 1, 31, 256, 3000])
def test_decode_scalar_batch(hip, n):
    fid = 0
    vals = ([0, pyref.P[fid] - 1, 1] + rand_values(fid, max(n, 3), 600 + n))[:n]
    frame = np.frombuffer(pyref.wire_frame("ScalarBatch", 42 + n, pyref.wire_scalar_records(fid, vals)), dtype=np.uint8).copy()
    out = np.zeros(4 * max(n, 1), dtype=np.uint64)
    cnt, rid = hip.eng(fid).wire_decode_scalar_batch(frame, len(frame), max(n, 1), out)
    assert (cnt, rid) == (n, 42 + n)
    assert np.array_equal(out[:4 * n], mont_array(fid, vals))


@pytest.mark.gpu
def test_point_batch_round_trip(hip, oracle):
    """PointBatch: compressed points from arkmpc_g1_to_bytes -> frame == model(frame of oracle bytes) -> records back."""
    from test_gpu_curve import random_points
    n = 50
    pts, P = random_points(n, 700)
    recs = hip.g1_to_bytes(P)
    eng = hip.eng(0)
    cap = eng.wire_frame_bound(n)
    buf = np.zeros(cap, dtype=np.uint8)
    ln = eng.wire_encode_bytes32(1, 77, n, recs, buf, cap)
    want = pyref.wire_frame("PointBatch", 77, [pyref.g1_compress(p) for p in pts])
    assert buf[:ln].tobytes() == want
    back = np.zeros(32 * n, dtype=np.uint8)
    cnt, rid, kind = eng.wire_decode_bytes32(buf[:ln].copy(), ln, n, back)
    assert (cnt, rid, kind) == (n, 77, 1) and np.array_equal(back, recs)


def _mutations(good):
    body = good[8:]
    def fr(b): return struct.pack("<Q", len(b)) + b
    yield "length prefix", struct.pack("<Q", len(body) + 1) + body
    yield "whitespace inside a number", fr(body.replace(b",255,", b",2 55,", 1))
    yield "byte > 255", fr(body.replace(b"[[", b"[[256,", 1).replace(b",0]", b"]", 1))
    yield "leading zero", fr(body.replace(b",255,", b",0255,", 1))
    yield "31 numbers", fr(body.replace(b",255,", b",", 1))
    yield "33 numbers", fr(body.replace(b",255,", b",255,255,", 1))
    yield "missing separator", fr(body.replace(b"],[", b"][", 1))
    yield "trailing garbage in body", fr(body.replace(b"]]}}", b"],]}}", 1))
    yield "other variant", fr(body.replace(b"ScalarBatch", b"ScalarShare", 1))
    yield "trailer", fr(body[:-1])
    yield "negative", fr(body.replace(b",255,", b",-25,", 1))


@pytest.mark.gpu
def test_decode_rejects_malformed_frames(hip, pkg):
    fid = 0
    vals = [255 * 256 + 7, 5, (255 << 8) | (255 << 40)] + rand_values(fid, 300, 801)
    good = pyref.wire_frame("ScalarBatch", 3, pyref.wire_scalar_records(fid, vals))
    eng = hip.eng(fid)
    out = np.zeros(4 * len(vals), dtype=np.uint64)
    assert eng.wire_decode_scalar_batch(np.frombuffer(good, dtype=np.uint8).copy(), len(good), len(vals), out)[0] == len(vals)
    seen = 0
    for name, bad in _mutations(good):
        assert bad != good, name
        arr = np.frombuffer(bad, dtype=np.uint8).copy()
        with pytest.raises(pkg.ArkMpcError):
            eng.wire_decode_scalar_batch(arr, len(arr), len(vals), out)
        seen += 1
    assert seen == 11
    # scalar >= modulus: deserialize_uncompressed rejects it (scalar.rs:195-201)
    p = pyref.P[fid]
    over = pyref.wire_frame("ScalarBatch", 3, [int(p).to_bytes(32, "little")])
    arr = np.frombuffer(over, dtype=np.uint8).copy()
    with pytest.raises(pkg.ArkMpcError):
        eng.wire_decode_scalar_batch(arr, len(arr), 4, out)
    # output capacity too small
    with pytest.raises(pkg.ArkMpcError):
        eng.wire_decode_scalar_batch(np.frombuffer(good, dtype=np.uint8).copy(), len(good), 10, out)


@pytest.mark.gpu
def test_large_round_trip_device_buffers(pkg):
    """2^21 scalars (the d||e exchange of a 2^20-gate batch) on device buffers: decode(encode(x)) == x, the text checked
    against the model on a prefix and through its total length."""
    torch = pytest.importorskip("torch")
    n = 1 << 21
    e = pkg.Engine(0, device=0, stream=torch.cuda.current_stream().cuda_stream)
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    raw = torch.randint(-(2**63), 2**63 - 1, (4 * n,), dtype=torch.int64, device="cuda", generator=g)
    x = torch.empty_like(raw); e.scalar_from_canonical(n, raw, x)
    cap = e.wire_frame_bound(n)
    frame = torch.empty(cap, dtype=torch.uint8, device="cuda")
    ln = e.wire_encode_scalar_batch(99, n, x, frame, cap)
    canon = torch.empty_like(x); e.scalar_to_canonical(n, x, canon); torch.cuda.synchronize()
    cb = canon.cpu().numpy().view(np.uint8).reshape(n, 32)
    digits = np.where(cb >= 100, 3, np.where(cb >= 10, 2, 1)).sum()
    assert ln == 8 + len(b'{"result_id":99,"payload":{"ScalarBatch":[') + int(digits) + 31 * n + 2 * n + (n - 1) + 3
    k = 2000
    model = pyref.wire_frame("ScalarBatch", 99, [bytes(r) for r in cb[:k]])
    head = frame[:len(model) - 3].cpu().numpy().tobytes()
    assert head[8:] == model[8:-3]
    back = torch.empty_like(x)
    cnt, rid = e.wire_decode_scalar_batch(frame, ln, n, back)
    assert (cnt, rid) == (n, 99) and torch.equal(back, x)


import json
import re

_UINT = re.compile(r"(?:0|[1-9][0-9]*)\Z")


class _Obj(list):
    """a JSON object as its list of (key, value) pairs: duplicates and order stay visible"""


def _no_const(_s):
    raise ValueError("NaN / Infinity are not JSON")


def _strings_and_depth(v, depth=1):
    """(every string in v, deepest nesting) -- depth counts like the decoder: a field's value is level 1"""
    if isinstance(v, _Obj):
        out, deepest = [], depth
        for k, x in v:
            ss, d = _strings_and_depth(x, depth + 1)
            out += [k] + ss; deepest = max(deepest, d)
        return out, deepest
    if isinstance(v, list):
        out, deepest = [], depth
        for x in v:
            ss, d = _strings_and_depth(x, depth + 1)
            out += ss; deepest = max(deepest, d)
        return out, deepest
    return ([v] if isinstance(v, str) else []), depth


def serde_model(fid, body, variant="ScalarBatch"):
    """What serde_json::from_slice::<NetworkOutbound> makes of `body` (network.rs:33-60, quic.rs:233-251), restated with Python's json: the
    (result_id, [values]) of a `variant` message or None.  Fields in any order, unknown fields skipped, a known field twice = error, keys
    compared after unescaping, exactly one variant key in the payload, integers only where integers are expected, scalars canonical."""
    try:
        text = body.decode("utf-8")
        msg = json.loads(text, object_pairs_hook=_Obj, parse_int=lambda t: ("int", t), parse_float=lambda t: ("float", t), parse_constant=_no_const)
    except (ValueError, RecursionError):
        return None
    if isinstance(msg, list) and not isinstance(msg, _Obj):     # serde's derive also implements visit_seq: [result_id, payload], exactly two elements
        if len(msg) != 2:
            return None
        msg = _Obj([("result_id", msg[0]), ("payload", msg[1])])
    if not isinstance(msg, _Obj):
        return None
    strings, depth = _strings_and_depth(msg, 0)
    if depth > 128 or any(0xD800 <= ord(ch) <= 0xDFFF for st in strings for ch in st):       # serde_json: recursion limit, no lone surrogates
        return None
    rid = payload = None
    for k, v in msg:
        if k == "result_id":
            if rid is not None:
                return None
            rid = v
        elif k == "payload":
            if payload is not None:
                return None
            payload = v
    if rid is None or payload is None:
        return None
    if not (isinstance(rid, tuple) and rid[0] == "int" and _UINT.match(rid[1]) and int(rid[1]) < 1 << 64):
        return None
    if not (isinstance(payload, _Obj) and len(payload) == 1 and payload[0][0] == variant and isinstance(payload[0][1], list) and not isinstance(payload[0][1], _Obj)):
        return None
    vals = []
    for e in payload[0][1]:
        if not (isinstance(e, list) and not isinstance(e, _Obj) and len(e) == 32):
            return None
        bs = []
        for t in e:
            if not (isinstance(t, tuple) and t[0] == "int" and _UINT.match(t[1]) and int(t[1]) <= 255):
                return None
            bs.append(int(t[1]))
        v = int.from_bytes(bytes(bs), "little")
        if v >= pyref.P[fid]:
            return None
        vals.append(v)
    return int(rid[1]), vals


def strict_accepts(fid, body):
    return serde_model(fid, body) is not None


@pytest.mark.gpu
def test_decoder_agrees_with_strict_grammar_on_mutated_frames(hip, pkg):
    """400 random byte edits of valid frames (replace / insert / delete, biased to structural characters): the decoder accepts exactly the
    frames the serde model above accepts, and when it accepts, the values are the model's."""
    import random
    fid = 0
    rng = random.Random(20260928)
    eng = hip.eng(fid)
    out = np.zeros(4 * 64, dtype=np.uint64)
    accepted = 0
    alphabet = b'[],0123456789{}":- eE.'
    for trial in range(400):
        n = rng.choice([0, 1, 2, 3, 9])
        vals = [rng.choice([0, 1, 255, 256, pyref.P[fid] - 1, rng.randrange(pyref.P[fid])]) for _ in range(n)]
        body = bytearray(pyref.wire_frame("ScalarBatch", rng.choice([0, 7, 10 ** 19, 2 ** 64 - 1]), pyref.wire_scalar_records(fid, vals))[8:])
        for _ in range(rng.choice([0, 1, 1, 2])):
            pos = rng.randrange(len(body))
            kind = rng.randrange(3)
            if kind == 0: body[pos] = rng.choice(alphabet)
            elif kind == 1: body.insert(pos, rng.choice(alphabet))
            else: del body[pos]
        frame = struct.pack("<Q", len(body)) + bytes(body)
        arr = np.frombuffer(frame, dtype=np.uint8).copy()
        want = strict_accepts(fid, bytes(body))
        try:
            cnt, rid = eng.wire_decode_scalar_batch(arr, len(arr), 64, out)
            got = True
        except pkg.ArkMpcError:
            got = False
        assert got == want, (trial, bytes(body)[:120])
        if got:
            accepted += 1
            want_rid, want_vals = serde_model(fid, bytes(body))
            assert rid == want_rid and cnt == len(want_vals)
            assert np.array_equal(out[:4 * cnt], mont_array(fid, want_vals))
    assert 40 < accepted < 360          # the mutation mix exercises both outcomes


def _object_shapes(fid, vals, rid):
    """(name, text, is a message) for frames serde_json::from_slice reads differently from a strict compact-form parser"""
    compact = pyref.wire_frame("ScalarBatch", rid, pyref.wire_scalar_records(fid, vals))[8:]
    arr = compact[compact.index(b":[") + 1:-2]                       # the [[..],[..]] text
    R, P = b'"result_id":%d' % rid, b'"payload":{"ScalarBatch":' + arr + b"}"
    deep = lambda k: b"[" * k + b"]" * k
    yield "fields swapped", b"{" + P + b"," + R + b"}", True
    yield "unknown field first", b'{"version":3,' + R + b"," + P + b"}", True
    yield "unknown fields with nested values", b'{"a":{"b":[1,2,{"c":null}],"d":"x\\\"y\u00e9"},' + R + b',"z":[true,false,-1.5e-3],' + P + b',"tail":""}', True
    yield "unknown field twice", b'{"x":1,"x":2,' + R + b"," + P + b"}", True
    yield "escaped known key", b'{"result\u005fid":%d,' % rid + P + b"}", True
    yield "escaped variant key", b"{" + R + b',"payload":{"Scalar\u0042atch":' + arr + b"}}", True
    yield "whitespace everywhere + swapped", b" \n{ " + P.replace(b":", b" : ", 2) + b" ,\t" + R + b" }\r\n", True
    yield "unknown value nested 100 deep", b'{"deep":' + deep(100) + b"," + R + b"," + P + b"}", True
    yield "duplicate result_id", b"{" + R + b"," + R + b"," + P + b"}", False
    yield "duplicate payload", b"{" + P + b"," + R + b"," + P + b"}", False
    yield "duplicate via escape", b'{"result\u005fid":1,' + R + b"," + P + b"}", False
    yield "missing payload", b"{" + R + b',"payloads":{}}', False
    yield "missing result_id", b'{"result id":1,' + P + b"}", False
    yield "two variants", b"{" + R + b',"payload":{"ScalarBatch":' + arr + b',"PointBatch":[]}}', False
    yield "empty payload object", b"{" + R + b',"payload":{}}', False
    yield "payload not an object", b"{" + R + b',"payload":' + arr + b"}", False
    yield "unknown variant", b"{" + R + b',"payload":{"ScalarBatches":' + arr + b"}}", False
    yield "trailing characters", b"{" + R + b"," + P + b"} x", False
    yield "two messages", b"{" + R + b"," + P + b"}{}", False
    yield "float result_id", b'{"result_id":%d.0,' % rid + P + b"}", False
    yield "negative zero result_id", b'{"result_id":-0,' + P + b"}", False
    yield "string result_id", b'{"result_id":"%d",' % rid + P + b"}", False
    yield "lone surrogate in an unknown field", b'{"s":"\ud800",' + R + b"," + P + b"}", False
    yield "surrogate pair in an unknown field", b'{"s":"\ud83d\ude00",' + R + b"," + P + b"}", True
    yield "raw control character in a string", b'{"s":"a\x01b",' + R + b"," + P + b"}", False
    yield "invalid utf-8 in a string", b'{"s":"\xff",' + R + b"," + P + b"}", False
    yield "bad literal", b'{"s":nul,' + R + b"," + P + b"}", False
    yield "leading zero in an unknown number", b'{"s":01,' + R + b"," + P + b"}", False
    yield "unknown value nested 200 deep", b'{"deep":' + deep(200) + b"," + R + b"," + P + b"}", False
    yield "byte order mark", b"\xef\xbb\xbf{" + R + b"," + P + b"}", False
    yield "top-level array", b"[" + R[12:] + b"]", False
    pay = b'{"ScalarBatch":' + arr + b"}"
    yield "sequence form [result_id, payload]", b"[%d," % rid + pay + b"]", True
    yield "sequence form with whitespace", b" [ %d ,\n " % rid + pay.replace(b":", b" : ", 1) + b" ]\t", True
    yield "sequence of three", b"[%d," % rid + pay + b",null]", False
    yield "sequence in the wrong order", b"[" + pay + b",%d]" % rid, False
    yield "sequence with a trailing comma", b"[%d," % rid + pay + b",]", False
    yield "sequence then garbage", b"[%d," % rid + pay + b"]]", False
    yield "sequence with a float id", b"[%d.5," % rid + pay + b"]", False


@pytest.mark.gpu
@pytest.mark.parametrize("device_frame", [False, True])
def test_decoder_has_serde_object_semantics(hip, pkg, device_frame):
    """network/quic.rs:233-251 reads a message with serde_json::from_slice into a derived struct: field order is free, unknown fields are skipped,
    duplicates of known fields are errors, keys are unescaped before they are compared.  Every shape below is also run through the Python
    model, so the test pins model and decoder against each other as well as against the expectation written next to the shape."""
    fid, rid = 0, 41
    vals = [0, 1, pyref.P[fid] - 1, 255, 256, 10 ** 30]
    n = len(vals)
    if device_frame:
        import torch
        e = pkg.Engine(fid, device=0, stream=torch.cuda.current_stream().cuda_stream)
    else:
        e = hip.eng(fid)
    for name, text, is_msg in _object_shapes(fid, vals, rid):
        assert (serde_model(fid, text) is not None) == is_msg, name
        frame = struct.pack("<Q", len(text)) + text
        arr = np.frombuffer(frame, dtype=np.uint8).copy()
        out = np.zeros(4 * n, dtype=np.uint64)
        try:
            if device_frame:
                dfr = torch.from_numpy(arr).cuda(); dout = torch.zeros(4 * n, dtype=torch.int64, device="cuda")
                cnt, got_rid = e.wire_decode_scalar_batch(dfr, len(frame), n, dout)
                out = dout.cpu().numpy().view(np.uint64)
            else:
                cnt, got_rid = e.wire_decode_scalar_batch(arr, len(frame), n, out)
            ok = True
        except pkg.ArkMpcError:
            ok = False
        assert ok == is_msg, name
        if ok:
            assert (cnt, got_rid) == (n, rid) and np.array_equal(out, mont_array(fid, vals)), name
    if device_frame:
        e.close()


def test_serde_model_reads_the_compact_form():
    """the model itself against the literal known answer (CPU)"""
    v = 1 + 255 * 256 + (10 << 248)
    body = b'{"result_id":7,"payload":{"ScalarBatch":[[1,255' + b",0" * 29 + b",10]]}}"
    assert serde_model(0, body) == (7, [v % (1 << 256)]) or v >= pyref.P[0]
    assert serde_model(0, body.replace(b'"result_id":7,', b"") ) is None
    assert serde_model(0, b'{"payload":{"ScalarBatch":[]},"result_id":18446744073709551615}') == ((1 << 64) - 1, [])
    assert serde_model(0, b'{"payload":{"ScalarBatch":[]},"result_id":18446744073709551616}') is None


# ---- round 2: JSON whitespace is accepted as serde_json::from_slice accepts it; exact-size device frames are never over-read ----
def _with_ws(frame: bytes, mode: str) -> bytes:
    """re-space the JSON text of a frame (token-level whitespace only) and fix the length prefix"""
    text = frame[8:]
    if mode == "pretty":
        import json as _json
        text = _json.dumps(_json.loads(text), indent=2).encode()
    elif mode == "mixed":
        text = text.replace(b",", b" ,\t").replace(b":", b" : ").replace(b"[", b"[\r\n ").replace(b"]", b" ]") + b"\n"
    return struct.pack("<Q", len(text)) + text


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["pretty", "mixed"])
@pytest.mark.parametrize("device_frame", [False, True])
def test_decode_accepts_json_whitespace(hip, pkg, mode, device_frame):
    fid, n = 0, 70
    vals = [0, 1, pyref.P[fid] - 1, 255, 256] + rand_values(fid, n - 5, 811)
    frame = _with_ws(pyref.wire_frame("ScalarBatch", 99, pyref.wire_scalar_records(fid, vals)), mode)
    out = np.zeros(4 * n, dtype=np.uint64)
    if device_frame:
        import torch
        e = pkg.Engine(fid, device=0, stream=torch.cuda.current_stream().cuda_stream)
        dfr = torch.from_numpy(np.frombuffer(frame, dtype=np.uint8).copy()).cuda()
        dout = torch.zeros(4 * n, dtype=torch.int64, device="cuda")
        cnt, rid = e.wire_decode_scalar_batch(dfr, len(frame), n, dout)
        out = dout.cpu().numpy().view(np.uint64)
        e.close()
    else:
        cnt, rid = hip.eng(fid).wire_decode_scalar_batch(np.frombuffer(frame, dtype=np.uint8).copy(), len(frame), n, out)
    assert (cnt, rid) == (n, 99) and np.array_equal(out, mont_array(fid, vals))


@pytest.mark.gpu
def test_whitespace_inside_tokens_is_rejected(hip, pkg):
    """`1 2` is two numbers without a separator and `"result_ id"` is another key: both invalid for serde_json as well"""
    e = hip.eng(0)
    good = pyref.wire_frame("ScalarBatch", 5, pyref.wire_scalar_records(0, [300, 7]))
    for old, new in ((b"[44,1,", b"[4 4,1,"), (b'"result_id"', b'"result_ id"'), (b'"ScalarBatch"', b'"Scalar Batch"')):
        assert old in good
        text = good[8:].replace(old, new, 1)
        frame = struct.pack("<Q", len(text)) + text
        with pytest.raises(pkg.ArkMpcError):
            e.wire_decode_scalar_batch(np.frombuffer(frame, dtype=np.uint8).copy(), len(frame), 2, np.zeros(8, dtype=np.uint64))


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 2, 3, 17, 256, 257])
def test_device_frame_of_exact_size_is_not_over_read(pkg, n):
    """Device mode with a frame allocation of EXACTLY frame_len bytes placed at the end of a larger buffer whose tail is filled with
    '[' bytes: a decoder that looked past frame_len would count phantom elements (and, on an exact allocation, read out of bounds)."""
    import torch
    fid = 0
    e = pkg.Engine(fid, device=0, stream=torch.cuda.current_stream().cuda_stream)
    vals = rand_values(fid, n, 900 + n)
    frame = pyref.wire_frame("ScalarBatch", n, pyref.wire_scalar_records(fid, vals))
    buf = torch.full((len(frame) + 64,), ord("["), dtype=torch.uint8, device="cuda")
    buf[:len(frame)] = torch.from_numpy(np.frombuffer(frame, dtype=np.uint8).copy()).cuda()
    out = torch.zeros(4 * (n + 8), dtype=torch.int64, device="cuda")
    cnt, rid = e.wire_decode_scalar_batch(buf, len(frame), n + 8, out)
    assert (cnt, rid) == (n, n)
    assert np.array_equal(out.cpu().numpy().view(np.uint64)[:4 * n], mont_array(fid, vals))
    e.close()
