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


# ---- round 2: the authenticated-point protocol pieces on Curve25519 (authenticated_curve.rs:66-283, 682-806 are generic over C) ----
@pytest.mark.parametrize("party", [0, 1])
def test_edshare_sub_public_and_scalarshare_mul_point(eng, oracle, party):
    n = 9
    _, A = rand_points(2 * n, 15)
    _, PUB = rand_points(n, 17)
    key = mont_array(2, rand_values(2, 1, 19))
    SS = mont_array(2, [0, 1] + rand_values(2, 2 * n - 2, 20))
    o = np.zeros(32 * n, dtype=np.uint64)
    eng.edshare_sub_public(n, party, key, A, PUB, o)                    # curve/share.rs:63-65
    assert aff_equal(eng, oracle, o, oracle.edshare_sub_public(party, key, A, PUB))
    assert aff_equal(eng, oracle, o, oracle.edshare_add_public(party, key, A, oracle.ed_batch_neg(PUB)))
    eng.scalarshare_mul_ed_point(n, SS, PUB, o)                         # curve.rs:483-517: point j = PUB[j // 2] * SS[j]
    assert aff_equal(eng, oracle, o, oracle.ed_batch_scalar_mul(PUB, SS, n=2 * n, p_div=2))


def test_ed_open_mac_check_commit_verify(eng, oracle):
    """open_authenticated_batch's local steps on Curve25519: `.share()` extraction, value*key - mac, per-element commitments over
    the compressed encoding, my + peer == identity -- against the oracle, hashlib, and a corrupted MAC."""
    import hashlib
    n = 12
    k0, k1 = rand_values(2, 2, 31)
    key = (k0 + k1) % pyref.EL
    vals = rand_values(2, n, 32)
    s0 = rand_values(2, n, 33); s1 = [(v - a) % pyref.EL for v, a in zip(vals, s0)]
    m0 = rand_values(2, n, 34); m1 = [(key * v - a) % pyref.EL for v, a in zip(vals, m0)]
    G = ext([pyref.ED_B], [1])
    mk = lambda ks, seed: ext([pyref.ed_mul(pyref.ED_B, k) for k in ks], [2 + (seed + 3 * i) % 89 for i in range(len(ks))])
    sh = []
    for sv, mv, seed in ((s0, m0, 1), (s1, m1, 2)):
        S, M = mk(sv, seed).reshape(n, 16), mk(mv, seed + 7).reshape(n, 16)
        sh.append(np.ascontiguousarray(np.concatenate([S, M], axis=1).reshape(-1)))
    keys = [mont_array(2, [k0]), mont_array(2, [k1])]
    mine = []
    for p in (0, 1):
        o = np.zeros(16 * n, dtype=np.uint64); eng.edshare_extract(n, sh[p], o)
        assert np.array_equal(o, np.ascontiguousarray(sh[p].reshape(n, 32)[:, :16]).reshape(-1))
        mine.append(o)
    opened = np.zeros(16 * n, dtype=np.uint64); eng.ed_add(n, mine[0], mine[1], opened)
    assert aff_equal(eng, oracle, opened, mk(vals, 5))                 # open == value * G
    chk = []
    for p in (0, 1):
        c = np.zeros(16 * n, dtype=np.uint64); eng.ed_mac_check_shares(n, keys[p], opened, sh[p], c)
        assert aff_equal(eng, oracle, c, oracle.ed_mac_check_shares(keys[p], opened, sh[p]))
        chk.append(c)
    ok = np.zeros(n, dtype=np.uint8); eng.ed_mac_verify(n, chk[0], chk[1], ok)
    assert ok.all() and all(oracle.ed_is_identity_sum(chk[0][16 * i:16 * i + 16], chk[1][16 * i:16 * i + 16]) for i in range(n))
    # commitments: SHA3-256(compressed point || BE(blinder)) reduced mod l
    bl = rand_values(2, n, 40); BL = mont_array(2, bl)
    comm = np.zeros(4 * n, dtype=np.uint64); eng.commit_ed_points_sha3(n, chk[0], BL, comm)
    data = np.zeros(32 * n, dtype=np.uint8); eng.ed_to_bytes(n, chk[0], data)
    assert np.array_equal(data, oracle.ed_to_bytes(chk[0]))
    for i in range(n):
        msg = data[32 * i:32 * i + 32].tobytes() + int(bl[i]).to_bytes(32, "big")
        want = int.from_bytes(hashlib.sha3_256(msg).digest(), "big") % pyref.EL
        assert pyref.from_mont(2, limbs_to_ints(comm[4 * i:4 * i + 4])[0]) == want
        assert np.array_equal(comm[4 * i:4 * i + 4], oracle.commit_bytes(2, data[32 * i:32 * i + 32].tobytes(), BL[4 * i:4 * i + 4].copy()))
    # a corrupted MAC share is caught at exactly that element (authenticated_curve.rs:1127-1155 modify_mac)
    bad = sh[0].copy(); bad[32 * 3 + 16: 32 * 3 + 32] = ext([pyref.ed_mul(pyref.ED_B, 12345)], [1])
    c = np.zeros(16 * n, dtype=np.uint64); eng.ed_mac_check_shares(n, keys[0], opened, bad, c)
    eng.ed_mac_verify(n, c, chk[1], ok)
    assert ok.tolist() == [1, 1, 1, 0] + [1] * (n - 4)


def test_ed_sums(eng, oracle):
    for n in (0, 1, 7, 300):
        _, A = rand_points(2 * n, 60 + n) if n else (None, np.zeros(0, dtype=np.uint64))
        o = np.zeros(32, dtype=np.uint64); eng.edshare_sum(n, A, o)
        want = np.concatenate([oracle.ed_sum(A, 32, 0), oracle.ed_sum(A, 32, 16)]) if n else np.concatenate([oracle.ed_identity()] * 2)
        assert aff_equal(eng, oracle, o, want)
        o1 = np.zeros(16, dtype=np.uint64); eng.ed_sum(2 * n, A, o1)
        assert aff_equal(eng, oracle, o1, oracle.ed_sum(A) if n else oracle.ed_identity())


def test_edshare_mul_public_at_scale_vs_oracle(pkg, oracle):
    """The hand-scheduled Curve25519 window loop on 2^13 PointShares (2^14 scalar-muls; ARKMPC_SOAK=full: 2^15) against the oracle's
    double-and-add on affine coordinates: random scalars plus scalars built so that every window sees every digit of the signed recoding
    (0, +-1 ... +-15, +16, carries into the top window), points with random Z.  Hosts with fewer than 16 cores check a 2^11 sample."""
    import os
    import torch
    n = 1 << (15 if os.environ.get("ARKMPC_SOAK") == "full" else 13)
    e = pkg.Engine("curve25519_fr", device=0, stream=torch.cuda.current_stream().cuda_stream)
    g = torch.Generator(device="cuda"); g.manual_seed(0xED25519)
    def rnd(cnt):
        raw = torch.randint(-(2**63), 2**63 - 1, (4 * cnt,), dtype=torch.int64, device="cuda", generator=g)
        out = torch.empty_like(raw); e.scalar_from_canonical(cnt, raw, out); return out
    k = rnd(2 * n)
    shares = torch.empty(32 * n, dtype=torch.int64, device="cuda")
    e.scalarshare_mul_ed_generator(n, k, shares)                 # P_i = k_i B: extended points with Z = 1 from the fixed-base path ...
    # ... and doubled points (Z != 1) for the other half of the batch
    dbl = torch.empty_like(shares); e.edshare_add(n, shares, shares, dbl)
    pts = torch.cat([shares[:16 * n], dbl[16 * n:]])
    L = pyref.EL
    special = []
    for d in list(range(0, 32)):
        special.append(sum(d << (5 * j) for j in range(51)) % L)                       # the same 5-bit pattern in every window
    special += [0, 1, 2, 15, 16, 17, 31, 32, L - 1, L - 2, (1 << 252), (1 << 252) - 1, (L - 1) // 2]
    sc_host = mont_array(2, special + rand_values(2, n - len(special), 777))
    sc = torch.from_numpy(sc_host.view(np.int64)).cuda()
    out = torch.empty_like(pts)
    e.edshare_mul_public(n, pts, sc, out)
    xy = torch.empty(8 * 2 * n, dtype=torch.int64, device="cuda"); e.ed_to_affine(2 * n, out, xy)
    torch.cuda.synchronize()
    threads = oracle.host_threads()
    idx = np.arange(n) if threads >= 16 else np.concatenate([np.arange(64), np.arange(64, n, max(1, n >> 11))])
    hp = np.ascontiguousarray(pts.cpu().numpy().view(np.uint64).reshape(n, 32)[idx]).reshape(-1)
    hs = np.ascontiguousarray(sc_host.reshape(n, 4)[idx]).reshape(-1)
    want = oracle.ed_batch_scalar_mul_mt(hp, hs, 2 * len(idx), p_div=1, s_div=2)
    wxy = oracle.ed_batch_to_affine_mt(want)
    gxy = xy.cpu().numpy().view(np.uint64).reshape(n, 16)[idx].reshape(-1)
    bad = np.nonzero((gxy.reshape(-1, 8) != wxy.reshape(-1, 8)).any(axis=1))[0]
    assert bad.size == 0, "%d of %d scalar-muls differ from the oracle, first at %d" % (bad.size, 2 * len(idx), bad[0])
    e.close()


def test_generator_mul_digit_edges(eng, oracle):
    """The fixed-base chain (signed 11-bit digits, 23 windows, plain-arithmetic asm additions): scalars that put every digit at an edge --
    0, +-1, +-1024 (the recoding threshold), 1025 in the top window (group order - 1 and its neighbours carry into it), all-ones windows,
    single digits in every window position -- against the oracle's double-and-add on the base point."""
    l = pyref.EL
    ks = [0, 1, 2, l - 1, l - 2, l - 1024, l - 1025, 1 << 252, (1 << 252) + 1, (1 << 252) - 1, (1 << 253) % l, ((1 << 253) - 1) % l]
    for w in range(23):
        for d in (1, 1023, 1024, 1025, 2047):
            ks.append((d << (11 * w)) % l)
            ks.append((l - (d << (11 * w))) % l)
    ks.append(sum(1024 << (11 * w) for w in range(22)))            # every window at the threshold
    ks.append(sum(1025 << (11 * w) for w in range(22)))            # every window just above it: a chain of carries
    ks += rand_values(2, 40, 4242)
    n = len(ks)
    S = mont_array(2, ks)
    o = np.zeros(16 * n, dtype=np.uint64); eng.ed_generator_mul(n, S, o)
    xa = np.zeros(8 * n, dtype=np.uint64); eng.ed_to_affine(n, o, xa)
    for i in range(n):
        x, y = limbs_to_ints(xa[8 * i:8 * i + 8])
        assert (pyref.from_mont(4, x), pyref.from_mont(4, y)) == pyref.ed_mul(pyref.ED_B, ks[i]), hex(ks[i])
    # the extended representative must be consistent too: T Z == X Y
    G = ext([pyref.ED_B] * n, [1] * n)
    assert aff_equal(eng, oracle, o, oracle.ed_batch_scalar_mul(G, S))


@pytest.mark.parametrize("party", [0, 1])
def test_edpoint_beaver_finish_equals_the_references_four_terms(eng, oracle, party):
    """arkmpc_edpoint_beaver_finish (the point-side K3 on Curve25519, regrouped to ([a] + d) eG + ([c] + d[b]) G) against the reference's literal
    terms deG + d[bG] + [a]eG + [c]G (authenticated_curve.rs:703-713) evaluated with the oracle's per-op functions, on affine coordinates."""
    n = 13
    l = pyref.EL
    key = mont_array(2, rand_values(2, 1, 700 + party))
    d = mont_array(2, [0, 1, l - 1] + rand_values(2, n - 3, 701))
    _, eG = rand_points(n, 702)                                                                 # includes the identity
    ta, tb, tc = (mont_array(2, rand_values(2, 2 * n, 703 + k)) for k in range(3))             # n ScalarShares each
    G = ext([pyref.ED_B], [1])
    bG = oracle.ed_batch_scalar_mul(G, tb, n=2 * n, p_div=2 * n)                                # [b]G     as n EdPointShares
    cG = oracle.ed_batch_scalar_mul(G, tc, n=2 * n, p_div=2 * n)                                # [c]G
    deG = oracle.ed_batch_scalar_mul(eG, d)                                                     # d * eG
    dbG = oracle.ed_batch_scalar_mul(bG, d, n=2 * n, s_div=2)                                   # d [bG]
    aeG = oracle.ed_batch_scalar_mul(eG, ta, n=2 * n, p_div=2)                                  # [a] eG
    want = oracle.ed_batch_add(oracle.edshare_add_public(party, key, dbG, deG), oracle.ed_batch_add(aeG, cG))
    out = np.zeros(32 * n, dtype=np.uint64)
    eng.point_beaver_finish(n, party, key, d, eG, ta, tb, tc, out, ed=True)
    assert aff_equal(eng, oracle, out, want)


# ---- variable-base MSM on Curve25519 (arkmpc_ed_msm / arkmpc_ed_msm_authenticated): CurvePoint::msm is generic over C (curve.rs:549-560)
def _ed_msm(eng, P, S):
    n = len(P) // 16
    out = np.zeros(16, dtype=np.uint64)
    eng.ed_msm(n, P if n else np.zeros(16, dtype=np.uint64), S if n else np.zeros(4, dtype=np.uint64), out)
    return out


@pytest.mark.parametrize("n", [0, 1, 2, 7, 33, 200])
def test_ed_msm_small_vs_oracle(eng, oracle, n):
    """the bucket method vs the definition (sum of scalar multiples), edge scalars 0, 1, l - 1 and the identity among the bases"""
    pts, P = rand_points(max(n, 1), 1900 + n)
    P = np.ascontiguousarray(P[:16 * n])
    ks = ([0, 1, pyref.EL - 1, 2, (1 << 252)] + rand_values(2, max(n, 5), 1901 + n))[:n]
    S = mont_array(2, ks)
    want = oracle.ed_msm(P, S) if n else np.array(pyref.ed_extended_mont((0, 1)), dtype=np.uint64)
    assert aff_equal(eng, oracle, _ed_msm(eng, P, S), want)


def test_ed_msm_repeated_and_cancelling_members(eng, oracle):
    """a base repeated with equal scalars (doubling inside a bucket), a base and its negative (cancellation inside a bucket), another
    representative of the same point: the addition law is complete, so these are ordinary lanes -- no flags, no fallback"""
    n = 16
    pts, P = rand_points(n, 1950)
    ks = rand_values(2, n, 1951)
    P[16 * 3:16 * 4] = P[16 * 2:16 * 3]; ks[3] = ks[2]
    P[16 * 5:16 * 6] = oracle.ed_batch_neg(P[16 * 4:16 * 5].copy()); ks[5] = ks[4]
    P[16 * 7:16 * 8] = ext([pts[6]], [12345]); ks[7] = ks[6]
    ks[8] = 0
    S = mont_array(2, ks)
    assert aff_equal(eng, oracle, _ed_msm(eng, P, S), oracle.ed_msm(P, S))
    Q = np.concatenate([P[:32], oracle.ed_batch_neg(P[:32].copy())])               # everything cancels
    S2 = mont_array(2, ks[:2] + ks[:2])
    assert aff_equal(eng, oracle, _ed_msm(eng, Q, S2), np.array(pyref.ed_extended_mont((0, 1)), dtype=np.uint64))


@pytest.mark.parametrize("n", [1000, 70000])
def test_ed_msm_closed_form_and_authenticated(eng, oracle, n):
    """size-independent check: bases k_i * B => msm = (sum s_i k_i mod l) * B, evaluated in Python integers (RFC 8032 base point);
    the authenticated form gives that per column (share column, MAC column)"""
    l = pyref.EL
    ks, s0, s1 = rand_values(2, n, 2001), rand_values(2, n, 2002), rand_values(2, n, 2003)
    ks[0], s0[0], s1[1] = 0, 5, 0                                                   # an identity base, a zero scalar
    P = np.zeros(16 * n, dtype=np.uint64)
    eng.ed_generator_mul(n, mont_array(2, ks), P)
    want0 = pyref.ed_mul(pyref.ED_B, sum(a * b for a, b in zip(ks, s0)) % l)
    want1 = pyref.ed_mul(pyref.ED_B, sum(a * b for a, b in zip(ks, s1)) % l)
    got = _ed_msm(eng, P, mont_array(2, s0))
    xy = np.zeros(8, dtype=np.uint64); eng.ed_to_affine(1, got, xy)
    assert [pyref.from_mont(4, v) for v in limbs_to_ints(xy)] == list(want0)
    ss = np.ascontiguousarray(np.concatenate([mont_array(2, s0).reshape(-1, 4), mont_array(2, s1).reshape(-1, 4)], axis=1).reshape(-1))
    out = np.zeros(32, dtype=np.uint64)
    eng.ed_msm_authenticated(n, P, ss, out)
    xy2 = np.zeros(16, dtype=np.uint64); eng.ed_to_affine(2, out, xy2)
    assert [pyref.from_mont(4, v) for v in limbs_to_ints(xy2)] == list(want0) + list(want1)


def test_ed_msm_skewed_scalars(eng, oracle):
    """all scalars equal / 0-1 scalars: one bucket per window holds every point (long runs cut into tasks and re-joined by the tree)"""
    n = 5000
    l = pyref.EL
    ks = rand_values(2, n, 2101)
    P = np.zeros(16 * n, dtype=np.uint64)
    eng.ed_generator_mul(n, mont_array(2, ks), P)
    for sc in ([7] * n, [i & 1 for i in range(n)], [l - 1] * n):
        want = pyref.ed_mul(pyref.ED_B, sum(a * b for a, b in zip(ks, sc)) % l)
        got = _ed_msm(eng, P, mont_array(2, sc))
        xy = np.zeros(8, dtype=np.uint64); eng.ed_to_affine(1, got, xy)
        assert [pyref.from_mont(4, v) for v in limbs_to_ints(xy)] == list(want)


def test_unsaturated_field_arithmetic_at_the_edges(pkg):
    """arkmpc_test_f9 (a test hook of the library, not in the header): the 29-bit-limb plain arithmetic of the Curve25519 MSM kernels on 256-bit
    integers chosen at the edges of its bounds -- against Python integers mod 2^255 - 19."""
    import ctypes
    import random
    q = (1 << 255) - 19
    edge = [0, 1, 2, 19, q - 1, q, q + 1, (1 << 255) - 20, (1 << 255) - 1, 1 << 255, (1 << 256) - 1, (1 << 256) - 38, (1 << 252), (1 << 29) - 1, ((1 << 232) - 1),
            int("1" * 29 + "0" * 29 + "1" * 29 + "0" * 29 + "1" * 29 + "0" * 29 + "1" * 29 + "0" * 29 + "1" * 24, 2) % (1 << 256), sum(((1 << 29) - 1) << (29 * i) for i in range(8)) | (0xffffff << 232)]
    rng = random.Random(5)
    A = [x for x in edge for _ in edge] + [rng.randrange(1 << 256) for _ in range(400)]
    B = [y for _ in edge for y in edge] + [rng.randrange(1 << 256) for _ in range(400)]
    n = len(A)
    from helpers import ints_to_limbs, limbs_to_ints
    e = pkg.Engine("curve25519_fr", device=0, host_buffers=True)
    a, b = ints_to_limbs(A), ints_to_limbs(B)
    out = np.zeros(20 * n, dtype=np.uint64)
    rc = e.lib.arkmpc_test_f9(e.h, ctypes.c_size_t(n), ctypes.c_void_p(a.ctypes.data), ctypes.c_void_p(b.ctypes.data), ctypes.c_void_p(out.ctypes.data))
    assert rc == 0
    got = np.array(limbs_to_ints(out), dtype=object).reshape(n, 5)
    for i, (x, y) in enumerate(zip(A, B)):
        want = [x * y % q, (x + y) % q, (x - y) % q, (x - y) * (x + y) % q, pow(x, q - 2, q)]
        assert list(got[i]) == want, (hex(x), hex(y))
    e.close()


def test_both_limb_forms_of_the_edwards_kernels_agree_with_the_oracle():
    """The 32-bit-limb window loop / table / fixed-base chain stay selectable (ARKMPC_ED_LIMBS=32; default: nine 29-bit limbs).  The same batch --
    identity, small and extreme scalars, random lanes, an EdPointShare x Scalar batch, generator multiples at the digit edges -- must equal the
    oracle in both forms, and the canonical affine outputs must be the same bytes."""
    import os
    import subprocess
    import sys
    import textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    prog = textwrap.dedent("""
        import importlib, sys, os, hashlib
        sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
        import numpy as np, pyref, oracle_api
        from helpers import mont_array, rand_values
        import test_gpu_edwards as T
        pkg = importlib.import_module("ark-mpc_amd"); eng = pkg.Engine("curve25519_fr", device=0, host_buffers=True); ora = oracle_api.load()
        n = 200
        pts, P = T.rand_points(n, 31)
        ks = [0, 1, 2, pyref.EL - 1, 15, 16, 17, 31, 32, 33, (1 << 252), pyref.EL - 2] + rand_values(2, n - 12, 32)
        S = mont_array(2, ks)
        o = np.zeros(16 * n, dtype=np.uint64); eng.ed_scalar_mul(n, P, S, o)
        ok1 = T.aff_equal(eng, ora, o, ora.ed_batch_scalar_mul(P, S))
        o2 = np.zeros(16 * n, dtype=np.uint64); eng.edshare_mul_public(n // 2, P, S[:4 * (n // 2)], o2)
        ok2 = T.aff_equal(eng, ora, o2, ora.ed_batch_scalar_mul(P, S[:4 * (n // 2)], n=n, s_div=2))
        edge = [0, 1, 1023, 1024, 1025, 2047, 2048, 2049, (1 << 11) - 1, (1 << 22) + 1024, pyref.EL - 1] + rand_values(2, 53, 33)
        SG = mont_array(2, edge); o3 = np.zeros(16 * len(edge), dtype=np.uint64); eng.ed_generator_mul(len(edge), SG, o3)
        ok3 = T.aff_equal(eng, ora, o3, ora.ed_batch_scalar_mul(T.ext([pyref.ED_B] * len(edge), [1] * len(edge)), SG))
        h = hashlib.sha256()
        for arr in (o, o2, o3):
            xa = np.zeros(len(arr) // 2, dtype=np.uint64); eng.ed_to_affine(len(arr) // 16, arr, xa); h.update(xa.tobytes())
        print(int(ok1), int(ok2), int(ok3), h.hexdigest())
    """ % (root, root))
    outs = {}
    for limbs in ("29", "32"):
        r = subprocess.run([sys.executable, "-c", prog], capture_output=True, text=True, env=dict(os.environ, ARKMPC_ED_LIMBS=limbs), timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[limbs] = r.stdout.strip().splitlines()[-1].split()
        assert outs[limbs][:3] == ["1", "1", "1"], (limbs, outs[limbs])
    assert outs["29"][3] == outs["32"][3]
