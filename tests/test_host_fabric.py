"""Protocol-level parity for the C++ host mirror (ark-mpc_amd/host/fabric.hpp) driving the HIP engine through the C ABI:
two parties over the in-process mock network with the reference's PartyIDBeaverSource, as the reference's own tests
run (lib.rs:116-128 execute_mock_mpc).  Expected values are exact integer arithmetic.
Includes BASELINE.json config 1: 1024 authenticated muls over Curve25519 Fr with the dummy Beaver source."""
import os
import struct
import subprocess

import numpy as np
import pytest

import pyref
from helpers import mixed_values, rand_values, ints_to_limbs, limbs_to_ints

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "ark-mpc_amd", "lib", "arkmpc_mock_mpc")


# Every scenario runs over every link mode and in both HBM layouts of the mirror's AuthenticatedScalarBatch -- as a covering set of three
# (link, layout) pairs by default; ARKMPC_SOAK=full runs the whole 3 x 2 product (each case is its own process: ~2 s of runtime start-up).
_PAIRS = [("host", "split"), ("device", "aos"), ("wire", "split")]
if os.environ.get("ARKMPC_SOAK") == "full":
    _PAIRS = [(l, y) for l in ("host", "device", "wire") for y in ("split", "aos")]


@pytest.fixture(params=_PAIRS, ids=["%s-%s" % p for p in _PAIRS], autouse=True)
def link_and_layout(request):
    """link: payloads handed over as host vectors (network/mock.rs), as device buffers (the same in-memory move for HBM-resident batches), or as
    the serde_json frames QuicTwoPartyNet carries (network/quic.rs:303-306), produced and parsed by the engine's wire codec on the GPU.
    layout: the engine-native split columns (the default: K1 reads no dead MAC bytes, the opening payload is the share column itself) or
    arkworks' AoS records."""
    os.environ["ARKMPC_MOCK_LINK"], os.environ["ARKMPC_SHARE_LAYOUT"] = request.param
    yield request.param
    os.environ.pop("ARKMPC_MOCK_LINK", None)
    os.environ.pop("ARKMPC_SHARE_LAYOUT", None)


def run(tmp_path, scenario, fid, a, b, *flags):
    n = len(a)
    inp, outp = tmp_path / "in.bin", tmp_path / "out.bin"
    inp.write_bytes(ints_to_limbs(a).tobytes() + ints_to_limbs(b).tobytes())
    r = subprocess.run([EXE, scenario, str(fid), str(n), str(inp), str(outp), *flags], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    frames = int(r.stdout.split("frames")[1].split()[0])
    assert (frames > 0) == (os.environ.get("ARKMPC_MOCK_LINK") == "wire"), r.stdout    # wire mode really framed the traffic
    raw = outp.read_bytes()
    res, off = [], 0
    for _ in range(2):
        err = struct.unpack_from("<Q", raw, off)[0]; off += 8
        vals = limbs_to_ints(np.frombuffer(raw, dtype=np.uint64, count=4 * n, offset=off)) if n else []
        off += 32 * n
        res.append((err, vals))
    return res


def test_driver_is_built():
    assert os.path.exists(EXE), "run __graft_entry__.build()"
    r = subprocess.run([EXE], capture_output=True, text=True)
    assert r.returncode == 2 and "usage" in r.stderr


@pytest.mark.gpu
def test_config1_curve25519_1024_muls_dummy_source(tmp_path):
    fid, n = 2, 1024
    p = pyref.P[fid]
    a, b = mixed_values(fid, n, 0xA11CE001), rand_values(fid, n, 0xA11CE001 + 1)
    res = run(tmp_path, "batch_mul", fid, a, b)
    want = [(x * y) % p for x, y in zip(a, b)]
    assert res[0] == (0, want) and res[1] == (0, want)       # MAC check passed on both sides, products correct


@pytest.mark.gpu
@pytest.mark.parametrize("fid", [0, 1])
@pytest.mark.parametrize("n", [1, 100])
def test_batch_mul_and_share_open(tmp_path, fid, n):
    p = pyref.P[fid]
    a, b = mixed_values(fid, n, 11), rand_values(fid, n, 12)
    res = run(tmp_path, "batch_mul", fid, a, b)
    want = [(x * y) % p for x, y in zip(a, b)]
    assert res[0] == (0, want) and res[1] == (0, want)
    res = run(tmp_path, "share_and_open", fid, a, b)
    assert res[0] == (0, a) and res[1] == (0, a)


@pytest.mark.gpu
def test_circuit(tmp_path):
    fid, n = 0, 257
    p = pyref.P[fid]
    a, b = rand_values(fid, n, 21), mixed_values(fid, n, 22)
    res = run(tmp_path, "circuit", fid, a, b)
    want = [(-(x * x - y * y) * x + x - y) % p for x, y in zip(a, b)]
    assert res[0] == (0, want) and res[1] == (0, want)


@pytest.mark.gpu
@pytest.mark.parametrize("fid,n", [(0, 1), (0, 300), (2, 77)])
def test_api_tail_pow_sum_constants(tmp_path, fid, n):
    """pow (authenticated_scalar.rs:86-100; pow(0) is the shared ZERO wire in the reference, pow(1) a clone), batch_add_constant (:531-560),
    Sum for AuthenticatedScalarResult (:563-575), ones_authenticated (fabric.rs:525-534), Sum / Product for ScalarResult
    (scalar_result.rs:325-338) and ScalarResult::batch_add_constant / batch_sub_constant, in one circuit with exact-integer expectations."""
    import functools
    p = pyref.P[fid]
    a, b = mixed_values(fid, n, 41), rand_values(fid, n, 42)
    res = run(tmp_path, "tail", fid, a, b)
    S = sum(pow(x, 5, p) + y for x, y in zip(a, b)) % p
    P = functools.reduce(lambda u, v: u * v % p, b, 1)
    T = sum(b) % p
    want = [(pow(x, 5, p) + y + S + 0 + 1 + P + T - y - x) % p for x, y in zip(a, b)]
    assert res[0] == (0, want) and res[1] == (0, want)


@pytest.mark.gpu
@pytest.mark.parametrize("devices,n", [("0,0,0", 1000), ("0,0,0,0,0,0,0,0", 5), ("0", 257)])
def test_group_fabric_one_party_over_several_gpus(tmp_path, devices, n):
    """host/fabric.hpp GroupFabric: each party drives an arkmpc_group (members sharing GPU 0 here); two dependent Beaver gates and the
    authenticated opening on range-sharded batches give (x*y)^2 on both sides, and a corrupted LAST share / MAC (the last member's range)
    is an AuthenticationError on both."""
    fid = 0
    p = pyref.P[fid]
    a, b = mixed_values(fid, n, 51), rand_values(fid, n, 52)
    os.environ["ARKMPC_GROUP_DEVICES"] = devices
    try:
        res = run(tmp_path, "group_mul", fid, a, b)
        want = [pow(x * y, 2, p) for x, y in zip(a, b)]
        assert res[0] == (0, want) and res[1] == (0, want)
        for flag in ("--bad-mac", "--bad-share"):
            res = run(tmp_path, "group_mul", fid, a, b, flag)
            assert res[0][0] == 2 and res[1][0] == 2
    finally:
        os.environ.pop("ARKMPC_GROUP_DEVICES", None)


@pytest.mark.gpu
@pytest.mark.parametrize("flag", ["--bad-mac", "--bad-share"])
def test_open_authenticated_detects_corruption(tmp_path, flag):
    """integration/src/authenticated_scalar.rs:49-75: a modified MAC or share must surface as AuthenticationError on
    BOTH parties (the honest one sees the sums fail, the corrupter too)."""
    fid, n = 0, 64
    a, b = rand_values(fid, n, 31), rand_values(fid, n, 32)
    res = run(tmp_path, "batch_mul", fid, a, b, flag)
    assert res[0][0] == 2 and res[1][0] == 2


@pytest.mark.gpu
def test_empty_batch(tmp_path):
    res = run(tmp_path, "batch_mul", 0, [], [])
    assert res[0] == (0, []) and res[1] == (0, [])


@pytest.mark.gpu
def test_point_beaver_mul_and_authenticated_open(tmp_path):
    """authenticated_curve.rs test_batch_mul (:1222-1244) restated: open(batch_mul([x], [y]G)) == (x*y) G for both parties,
    with the per-element commitment + MAC check passing; then one corrupted MAC point is caught for exactly that element."""
    fid, n = 0, 12
    r = pyref.RORD
    x, y = [0, 1, r - 1] + rand_values(fid, n - 3, 41), [5, 7, 2] + rand_values(fid, n - 3, 42)
    want = b"".join(pyref.g1_compress(pyref.g1_mul(pyref.G, (u * v) % r)) for u, v in zip(x, y))
    for flags, fails in (((), 0), (("--bad-mac",), 1)):
        inp, outp = tmp_path / "in.bin", tmp_path / "out.bin"
        inp.write_bytes(ints_to_limbs(x).tobytes() + ints_to_limbs(y).tobytes())
        rr = subprocess.run([EXE, "point_mul", str(fid), str(n), str(inp), str(outp), *flags], capture_output=True, text=True, timeout=600)
        assert rr.returncode == 0, rr.stderr
        raw = outp.read_bytes()
        off = 0
        for party in range(2):
            nfail = struct.unpack_from("<Q", raw, off)[0]; off += 8
            got = raw[off:off + 32 * n]; off += 32 * n
            assert nfail == fails
            assert got == want          # opening is unaffected by a corrupted MAC


@pytest.mark.gpu
@pytest.mark.parametrize("fid", [0, 2])
def test_batch_inverse_protocol(tmp_path, fid):
    """authenticated_scalar.rs:55-82: two-round inversion via a shared random mask; open(inverse(x)) == x^-1 mod p."""
    n = 65
    p = pyref.P[fid]
    a = [v for v in mixed_values(fid, n + 5, 51) if v != 0][:n]
    res = run(tmp_path, "inverse", fid, a, a)
    want = [pow(v, -1, p) for v in a]
    assert res[0] == (0, want) and res[1] == (0, want)


@pytest.mark.gpu
def test_authenticated_msm(tmp_path):
    """AuthenticatedPointResult::msm (authenticated_curve.rs:787-806): open(msm([x_i], [y_i]G)) == (sum x_i*y_i) G."""
    fid, n = 0, 40
    r = pyref.RORD
    x, y = rand_values(fid, n, 61), rand_values(fid, n, 62)
    want = pyref.g1_compress(pyref.g1_mul(pyref.G, sum(u * v for u, v in zip(x, y)) % r))
    inp, outp = tmp_path / "in.bin", tmp_path / "out.bin"
    inp.write_bytes(ints_to_limbs(x).tobytes() + ints_to_limbs(y).tobytes())
    rr = subprocess.run([EXE, "msm", str(fid), str(n), str(inp), str(outp)], capture_output=True, text=True, timeout=600)
    assert rr.returncode == 0, rr.stderr
    raw = outp.read_bytes()
    assert len(raw) == 2 * (8 + 32)
    for party in range(2):
        assert struct.unpack_from("<Q", raw, 40 * party)[0] == 0          # MAC check passed
        assert raw[40 * party + 8: 40 * party + 40] == want


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 300])
def test_msm_authenticated_public_points(tmp_path, n):
    """CurvePointResult::msm_authenticated (curve.rs:618-642): authenticated scalars x public bases [b_i]G through the
    bucket-method MSM; the authenticated open must give (sum x_i*b_i) G with the MAC check passing on both parties."""
    fid = 0
    r = pyref.RORD
    x, b = rand_values(fid, n, 63), rand_values(fid, n, 64)
    want = pyref.g1_compress(pyref.g1_mul(pyref.G, sum(u * v for u, v in zip(x, b)) % r))
    inp, outp = tmp_path / "in.bin", tmp_path / "out.bin"
    inp.write_bytes(ints_to_limbs(x).tobytes() + ints_to_limbs(b).tobytes())
    rr = subprocess.run([EXE, "msm_public_points", str(fid), str(n), str(inp), str(outp)], capture_output=True, text=True, timeout=600)
    assert rr.returncode == 0, rr.stderr
    raw = outp.read_bytes()
    assert len(raw) == 2 * (8 + 32)
    for party in range(2):
        assert struct.unpack_from("<Q", raw, 40 * party)[0] == 0
        assert raw[40 * party + 8: 40 * party + 40] == want


@pytest.mark.gpu
def test_batch_div_protocol(tmp_path):
    fid, n = 0, 33
    p = pyref.P[fid]
    a = rand_values(fid, n, 71)
    b = [v for v in mixed_values(fid, n + 5, 72) if v != 0][:n]
    res = run(tmp_path, "div", fid, a, b)
    want = [(x * pow(y, -1, p)) % p for x, y in zip(a, b)]
    assert res[0] == (0, want) and res[1] == (0, want)


@pytest.mark.gpu
def test_batch_share_point(tmp_path):
    """fabric.rs:622-649: input sharing of curve points (sender masks with mask*G), then authenticated open returns them."""
    fid, n = 0, 9
    a = [0, 1, pyref.RORD - 1] + rand_values(fid, n - 3, 81)
    want = b"".join(pyref.g1_compress(pyref.g1_mul(pyref.G, v)) for v in a)
    inp, outp = tmp_path / "in.bin", tmp_path / "out.bin"
    inp.write_bytes(ints_to_limbs(a).tobytes() + ints_to_limbs(a).tobytes())
    rr = subprocess.run([EXE, "share_point", str(fid), str(n), str(inp), str(outp)], capture_output=True, text=True, timeout=600)
    assert rr.returncode == 0, rr.stderr
    raw = outp.read_bytes()
    for party in range(2):
        off = party * (8 + 32 * n)
        assert struct.unpack_from("<Q", raw, off)[0] == 0
        assert raw[off + 8: off + 8 + 32 * n] == want


@pytest.mark.gpu
def test_xor_circuit(tmp_path):
    """test_xor_circuit (authenticated_scalar.rs:1677-1688) / gadgets.rs bit_xor_batch on all four bit pairs, batched."""
    fid = 0
    a = [0, 0, 1, 1] * 16
    b = [0, 1, 0, 1] * 16
    res = run(tmp_path, "xor", fid, a, b)
    want = [x ^ y for x, y in zip(a, b)]
    assert res[0] == (0, want) and res[1] == (0, want)


@pytest.mark.gpu
def test_prefix_product_gadget(tmp_path):
    """gadgets.rs:105-148: five Beaver/open rounds + a public scan; open(prefix_product(x))_i == x_0 * ... * x_i."""
    fid, n = 0, 300
    p = pyref.P[fid]
    a = [v for v in rand_values(fid, n, 91)]
    res = run(tmp_path, "prefix_product", fid, a, a)
    run_, want = 1, []
    for v in a:
        run_ = run_ * v % p; want.append(run_)
    assert res[0] == (0, want) and res[1] == (0, want)


# ---- round 2: the point-side protocols are curve-generic (BN254 G1 for field 0, Curve25519 for field 2, the reference's
# ---- README curve), sub_public, shared bits, and a peer that sends a short batch ----------------------------------------
def _curve(fid):
    if fid == 0:
        return pyref.RORD, (lambda k: pyref.g1_compress(pyref.g1_mul(pyref.G, k)))
    return pyref.EL, (lambda k: pyref.ed_compress(pyref.ed_mul(pyref.ED_B, k)))


def _run_points(tmp_path, scenario, fid, a, b, *flags):
    n = len(a)
    inp, outp = tmp_path / "in.bin", tmp_path / "out.bin"
    inp.write_bytes(ints_to_limbs(a).tobytes() + ints_to_limbs(b).tobytes())
    rr = subprocess.run([EXE, scenario, str(fid), str(n), str(inp), str(outp), *flags], capture_output=True, text=True, timeout=900)
    assert rr.returncode == 0, rr.stderr
    raw = outp.read_bytes()
    per = (len(raw) - 16) // 2
    return [(struct.unpack_from("<Q", raw, k * (8 + per))[0], raw[k * (8 + per) + 8: (k + 1) * (8 + per)]) for k in range(2)]


@pytest.mark.gpu
def test_curve25519_point_beaver_mul_and_authenticated_open(tmp_path):
    """AuthenticatedPointResult::batch_mul + open_authenticated_batch (authenticated_curve.rs:682-714, :190-283) over
    Curve25519 -- the curve of BASELINE config 1 / README.md:24: open == (x*y) B for both parties, MAC checks pass, and a
    corrupted MAC point is caught for exactly one element."""
    fid, n = 2, 10
    l, comp = _curve(fid)
    x, y = [0, 1, l - 1] + rand_values(fid, n - 3, 141), [5, 7, 2] + rand_values(fid, n - 3, 142)
    want = b"".join(comp((u * v) % l) for u, v in zip(x, y))
    for flags, fails in (((), 0), (("--bad-mac",), 1)):
        for nfail, got in _run_points(tmp_path, "point_mul", fid, x, y, *flags):
            assert nfail == fails and got == want


@pytest.mark.gpu
@pytest.mark.parametrize("fid", [0, 2])
def test_point_sub_public_and_share_point(tmp_path, fid):
    """PointShare::sub_public (curve/share.rs:63-65) through the mirror on both curves: share [a]G, subtract public [b]G,
    authenticated open gives [(a-b)]G; and batch_share_point round trip on Curve25519 (BN254 is covered above)."""
    n = 7
    l, comp = _curve(fid)
    a, b = [0, 1, l - 1] + rand_values(fid, n - 3, 151), [1, 1, 5] + rand_values(fid, n - 3, 152)
    want = b"".join(comp((u - v) % l) for u, v in zip(a, b))
    for nfail, got in _run_points(tmp_path, "point_sub_public", fid, a, b):
        assert nfail == 0 and got == want
    want = b"".join(comp(u) for u in a)
    for nfail, got in _run_points(tmp_path, "share_point", fid, a, a):
        assert nfail == 0 and got == want


@pytest.mark.gpu
def test_curve25519_authenticated_msm(tmp_path):
    """AuthenticatedPointResult::msm and CurvePointResult::msm_authenticated on Curve25519 (scalar-muls + point sums)."""
    fid, n = 2, 24
    l, comp = _curve(fid)
    x, y = rand_values(fid, n, 161), rand_values(fid, n, 162)
    want = comp(sum(u * v for u, v in zip(x, y)) % l)
    for scen in ("msm", "msm_public_points"):
        for nfail, got in _run_points(tmp_path, scen, fid, x, y):
            assert nfail == 0 and got == want


@pytest.mark.gpu
def test_random_shared_bits(tmp_path):
    """fabric.rs:961-984 over PreprocessingPhase::next_shared_bit_batch (offline_prep.rs:39-44): the dummy source hands out
    (party_id, party_id), so every bit opens to 0 + 1 = 1 under MAC key 1."""
    res = run(tmp_path, "shared_bits", 0, [0] * 33, [0] * 33)
    assert res[0] == (0, [1] * 33) and res[1] == (0, [1] * 33)


@pytest.mark.gpu
def test_short_peer_batch_is_a_network_error(tmp_path):
    """A peer whose batch is one element short must be rejected before any kernel is launched on it (the kernels run with the
    LOCAL n): both sides end with MpcNetworkError, no out-of-bounds read, no hang."""
    n = 50
    a = rand_values(0, n, 171)
    inp, outp = tmp_path / "in.bin", tmp_path / "out.bin"
    inp.write_bytes(ints_to_limbs(a).tobytes() + ints_to_limbs(a).tobytes())
    rr = subprocess.run([EXE, "short_peer", "0", str(n), str(inp), str(outp)], capture_output=True, text=True, timeout=300)
    assert rr.returncode == 1 and "MpcNetworkError" in rr.stderr, rr.stderr


@pytest.fixture
def dealer_source():
    """ARKMPC_MOCK_DEALER: the mirror's trusted-dealer source (random MAC key shares -- party 0's is 0 with the reference's dummy source --
    random triples, masks, bits and inverse pairs, random additive splits)."""
    os.environ["ARKMPC_MOCK_DEALER"] = "0x5EED0001"
    yield
    os.environ.pop("ARKMPC_MOCK_DEALER", None)


@pytest.mark.gpu
@pytest.mark.parametrize("fid", [0, 1, 2])
def test_scalar_protocols_with_random_preprocessing(tmp_path, dealer_source, fid):
    """batch_mul, a small circuit, batch_inverse and the corrupted-MAC / corrupted-share detections again, with non-degenerate
    preprocessing material: both parties' MAC key shares, every triple and every mask are random."""
    n = 130
    p = pyref.P[fid]
    a, b = mixed_values(fid, n, 801 + fid), rand_values(fid, n, 811 + fid)
    want = [(x * y) % p for x, y in zip(a, b)]
    res = run(tmp_path, "batch_mul", fid, a, b)
    assert res[0] == (0, want) and res[1] == (0, want)
    res = run(tmp_path, "batch_mul", fid, a, b, "--bad-mac")
    assert res[0][0] == 2 and res[1][0] == 2                    # AuthenticationError (out code 2) on BOTH parties
    res = run(tmp_path, "batch_mul", fid, a, b, "--bad-share")
    assert res[0][0] == 2 and res[1][0] == 2
    nz = [v for v in a if v != 0]
    res = run(tmp_path, "inverse", fid, nz, nz)
    want = [pow(v, -1, p) for v in nz]
    assert res[0] == (0, want) and res[1] == (0, want)


@pytest.mark.gpu
@pytest.mark.parametrize("fid", [0, 2])
@pytest.mark.parametrize("source", ["dummy", "dealer"])
def test_point_beaver_mul_regrouped_equals_literal_sequence(tmp_path, fid, source):
    """AuthenticatedPointResult::batch_mul (authenticated_curve.rs:682-714): the mirror evaluates deG + d[bG] + [a]eG + [c]G regrouped as
    ([a] + d) eG + ([c] + d[b]) G -- 2 variable-base + 4 generator scalar-muls instead of 6 + 4.  Scenario point_mul_forms runs both that
    form and the reference's literal op sequence on the same shares and counts the LOCAL share / MAC points that differ (as group elements,
    compared in the compressed encoding): none may, on either party, with the dummy source and with random preprocessing material; and
    the result opens to (x*y) G with every MAC check passing."""
    if source == "dealer":
        os.environ["ARKMPC_MOCK_DEALER"] = "0x5EED0002"
    try:
        n = 24
        l, comp = _curve(fid)
        x, y = [0, 1, l - 1, 5] + rand_values(fid, n - 4, 171), [5, 7, 2, 0] + rand_values(fid, n - 4, 172)
        want = b"".join(comp((u * v) % l) for u, v in zip(x, y))
        for nfail, got in _run_points(tmp_path, "point_mul_forms", fid, x, y):
            assert nfail == 0 and got == want                    # nfail carries 10^6 per differing local point + failed MAC checks
    finally:
        os.environ.pop("ARKMPC_MOCK_DEALER", None)


@pytest.mark.gpu
@pytest.mark.parametrize("fid", [0, 2])
def test_point_protocols_with_random_preprocessing(tmp_path, dealer_source, fid):
    """point batch_mul + authenticated open, sub_public, share_point and msm with random preprocessing; one corrupted MAC point is
    caught for exactly one element."""
    n = 9
    l, comp = _curve(fid)
    x, y = [0, 1, l - 1] + rand_values(fid, n - 3, 181), [5, 7, 2] + rand_values(fid, n - 3, 182)
    want = b"".join(comp((u * v) % l) for u, v in zip(x, y))
    for flags, fails in (((), 0), (("--bad-mac",), 1)):
        for nfail, got in _run_points(tmp_path, "point_mul", fid, x, y, *flags):
            assert nfail == fails and got == want
    want = b"".join(comp((u - v) % l) for u, v in zip(x, y))
    for nfail, got in _run_points(tmp_path, "point_sub_public", fid, x, y):
        assert nfail == 0 and got == want
    want = comp(sum(u * v for u, v in zip(x, y)) % l)
    for nfail, got in _run_points(tmp_path, "msm", fid, x, y):
        assert nfail == 0 and got == want


@pytest.mark.gpu
@pytest.mark.parametrize("prefetch", ["1", "0"])
@pytest.mark.parametrize("source", ["vector_lent_in_place", "dealer_vecs"])
def test_chain_of_gates_with_triples_read_ahead(tmp_path, dealer_source, prefetch, source):
    """Six dependent Beaver gates on resident operands, every gate on fresh random triples from a host-memory source (fabric.rs:894-915,
    offline_prep.rs:65-81) -- not the dummy source's one-record shortcut.  The triples go up asynchronously and, with prefetch on, one gate ahead
    of their use; a gate of another size (half, then full again) is served from what was read ahead, in FIFO order.  n is large enough for
    the vectors to be pinned in place and imported by the in-place kernel.  Both parties open 2 a b^5 with valid MACs."""
    fid, n = 0, 20000
    p = pyref.P[fid]
    a, b = mixed_values(fid, n, 901), rand_values(fid, n, 902)
    os.environ["ARKMPC_TRIPLE_PREFETCH"] = prefetch
    if source == "vector_lent_in_place":
        os.environ["ARKMPC_MOCK_VECTOR_TRIPLES"] = str(6 * n)
    try:
        res = run(tmp_path, "chain", fid, a, b)
        want = [(2 * x * pow(y, 5, p)) % p for x, y in zip(a, b)]
        assert res[0] == (0, want) and res[1] == (0, want)
        res = run(tmp_path, "chain", fid, a, b, "--bad-mac")
        assert res[0][0] == 2 and res[1][0] == 2
    finally:
        os.environ.pop("ARKMPC_TRIPLE_PREFETCH", None)
        os.environ.pop("ARKMPC_MOCK_VECTOR_TRIPLES", None)


@pytest.mark.gpu
def test_vector_source_running_out_is_an_error_not_a_short_batch(tmp_path, dealer_source):
    """LowGearPrep asserts when it runs out (offline-phase/src/structs.rs:189); reading ahead must not turn that into a silent short batch nor
    fire before the triples are really needed: 5.5 n are needed, 5.5 n - 1 are there"""
    fid, n = 0, 2000
    a, b = mixed_values(fid, n, 903), rand_values(fid, n, 904)
    inp, outp = tmp_path / "in.bin", tmp_path / "out.bin"
    inp.write_bytes(ints_to_limbs(a).tobytes() + ints_to_limbs(b).tobytes())
    for cap, ok in ((5 * n + n // 2, True), (5 * n + n // 2 - 1, False)):
        env = dict(os.environ, ARKMPC_MOCK_VECTOR_TRIPLES=str(cap))
        r = subprocess.run([EXE, "chain", str(fid), str(n), str(inp), str(outp)], capture_output=True, text=True, timeout=300, env=env)
        assert (r.returncode == 0) == ok, r.stderr
        if not ok:
            assert "preprocessing exhausted" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("devices,n", [("0,0,0", 20011), ("0,0,0,0,0,0,0,0", 5), ("0", 257)])
def test_group_fabric_host_vectors_through_group_sessions(tmp_path, dealer_source, devices, n):
    """GroupFabric::batch_mul_host: host records in, host records out through arkmpc_group_hostmul_* (one range session per member), two
    dependent gates, then the authenticated opening on the sharded path: (x y)^2 on both sides; a corrupted last element is detected"""
    fid = 0
    p = pyref.P[fid]
    a, b = mixed_values(fid, n, 905), rand_values(fid, n, 906)
    os.environ["ARKMPC_GROUP_DEVICES"] = devices
    try:
        res = run(tmp_path, "group_mul_host", fid, a, b)
        want = [pow(x * y, 2, p) for x, y in zip(a, b)]
        assert res[0] == (0, want) and res[1] == (0, want)
        res = run(tmp_path, "group_mul_host", fid, a, b, "--bad-share")
        assert res[0][0] == 2 and res[1][0] == 2
    finally:
        os.environ.pop("ARKMPC_GROUP_DEVICES", None)
