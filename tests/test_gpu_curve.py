"""GPU parity for the curve side (BN254 G1): HIP kernels through the C ABI vs the CPU oracle.
Points are compared on AFFINE coordinates (what arkworks' equality and wire format see); the Jacobian
representative depends on the addition chain."""
import numpy as np
import pytest

import pyref
from helpers import mont_array, rand_values, limbs_to_ints, EngineAdapter

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip(pkg):
    return EngineAdapter(pkg)


def jac(points, zs):
    return np.array(sum((pyref.g1_jacobian_mont(p, z) for p, z in zip(points, zs)), []), dtype=np.uint64)


def random_points(n, seed, with_identity=True):
    ks = rand_values(0, n, seed)
    pts = [pyref.g1_mul(pyref.G, k) for k in ks]
    if with_identity and n >= 4:
        pts[1] = None
    zs = [1 + (seed * 31 + 7 * i) % 997 for i in range(n)]
    return pts, jac(pts, zs)


def affine_equal(hip, oracle, a, b):
    xa, ia = hip.g1_batch_to_affine(np.ascontiguousarray(a))
    xb, ib = oracle.g1_batch_to_affine(np.ascontiguousarray(b))
    return np.array_equal(ia, ib) and np.array_equal(xa, xb)


def test_add_sub_neg(hip, oracle):
    n = 40
    _, P = random_points(n, 1)
    _, Q = random_points(n, 2)
    Q[12 * 5:12 * 6] = P[12 * 5:12 * 6]                                  # P + P through add
    Q[12 * 6:12 * 7] = oracle.g1_neg(P[12 * 6:12 * 7].copy())           # P + (-P)
    assert affine_equal(hip, oracle, hip.g1_batch_add(P, Q), oracle.g1_batch_add(P, Q))
    assert affine_equal(hip, oracle, hip.g1_neg(P), oracle.g1_neg(P))
    o = np.zeros(12 * n, dtype=np.uint64); hip.eng(0).g1_sub(n, P, Q, o)
    assert affine_equal(hip, oracle, o, oracle.g1_batch_add(P, oracle.g1_neg(Q)))
    # identity outputs are stored in arkworks' canonical form (1, 1, 0)
    s = hip.g1_batch_add(P[12 * 6:12 * 7].copy(), Q[12 * 6:12 * 7].copy())
    assert np.array_equal(s, oracle.g1_identity())


def test_scalar_mul(hip, oracle):
    n = 24
    pts, P = random_points(n, 3)
    ks = [0, 1, 2, pyref.RORD - 1] + rand_values(0, n - 4, 4)
    S = mont_array(0, ks)
    got = hip.g1_batch_scalar_mul(P, S)
    assert affine_equal(hip, oracle, got, oracle.g1_batch_scalar_mul(P, S))
    # and against the Python group law directly
    xy, inf = hip.g1_batch_to_affine(got)
    for i in range(n):
        want = pyref.g1_mul(pts[i], ks[i])
        if want is None:
            assert inf[i] == 1
        else:
            x, y = limbs_to_ints(xy[8 * i:8 * i + 8])
            assert (pyref.from_mont(3, x), pyref.from_mont(3, y)) == want
    o = np.zeros(12 * n, dtype=np.uint64); hip.eng(0).g1_generator_mul(n, S, o)
    G = jac([pyref.G] * n, [1] * n)
    assert affine_equal(hip, oracle, o, oracle.g1_batch_scalar_mul(G, S))
    assert np.array_equal(hip.g1_to_bytes(got), oracle.g1_to_bytes(oracle.g1_batch_scalar_mul(P, S)))


@pytest.mark.parametrize("party", [0, 1])
def test_pointshare_ops(hip, oracle, party):
    n = 10
    _, A = random_points(2 * n, 5)
    _, B = random_points(2 * n, 6)
    S = mont_array(0, [0, 1] + rand_values(0, n - 2, 7))
    key = mont_array(0, rand_values(0, 1, 8))
    _, PUB = random_points(n, 9)
    SS = mont_array(0, rand_values(0, 2 * n, 10))   # n ScalarShares = 2n scalars
    assert affine_equal(hip, oracle, hip.pointshare_add(A, B), oracle.pointshare_add(A, B))
    assert affine_equal(hip, oracle, hip.pointshare_add(A, B, sub=True), oracle.pointshare_add(A, B, sub=True))
    assert affine_equal(hip, oracle, hip.pointshare_neg(A), oracle.pointshare_neg(A))
    assert affine_equal(hip, oracle, hip.pointshare_mul_public(A, S), oracle.pointshare_mul_public(A, S))
    assert affine_equal(hip, oracle, hip.pointshare_add_public(party, key, A, PUB), oracle.pointshare_add_public(party, key, A, PUB))
    o = np.zeros(24 * n, dtype=np.uint64); hip.eng(0).pointshare_sub_public(n, party, key, A, PUB, o)      # curve/share.rs:63-65
    assert affine_equal(hip, oracle, o, oracle.pointshare_sub_public(party, key, A, PUB))
    assert affine_equal(hip, oracle, o, oracle.pointshare_add_public(party, key, A, oracle.g1_neg(PUB)))
    chk = np.zeros(12 * n, dtype=np.uint64); hip.eng(0).point_mac_check_shares(n, key, PUB, A, chk)         # authenticated_curve.rs:215-220
    assert affine_equal(hip, oracle, chk, oracle.point_mac_check_shares(key, PUB, A))
    assert affine_equal(hip, oracle, hip.scalarshare_mul_generator(SS), oracle.scalarshare_mul_generator(SS))
    assert affine_equal(hip, oracle, hip.scalarshare_mul_point(SS, PUB), oracle.scalarshare_mul_point(SS, PUB))
    o = np.zeros(12 * n, dtype=np.uint64); hip.eng(0).pointshare_extract(n, A, o)
    assert np.array_equal(o.reshape(-1, 12), A.reshape(-1, 24)[:, :12])


def test_point_open_authenticated(hip, oracle):
    """AuthenticatedPointResult::open_authenticated_batch (authenticated_curve.rs:190-283) for both parties:
    value*mac_key - mac per party sums to the identity; a corrupted MAC is caught (modify_mac :839-849)."""
    n = 6
    r = pyref.RORD
    k0, k1 = rand_values(0, 2, 11)
    key = (k0 + k1) % r
    vals = rand_values(0, n, 12)                     # discrete logs of the shared points
    s0 = rand_values(0, n, 13); s1 = [(v - a) % r for v, a in zip(vals, s0)]
    m0 = rand_values(0, n, 14); m1 = [(key * v - a) % r for v, a in zip(vals, m0)]
    mk = lambda ss, ms: jac(sum(([pyref.g1_mul(pyref.G, s), pyref.g1_mul(pyref.G, m)] for s, m in zip(ss, ms)), []), [1 + i for i in range(2 * n)])
    sh = [mk(s0, m0), mk(s1, m1)]
    keys = [mont_array(0, [k0]), mont_array(0, [k1])]
    e = hip.eng(0)
    mine = []
    for p in (0, 1):
        o = np.zeros(12 * n, dtype=np.uint64); e.pointshare_extract(n, sh[p], o); mine.append(o)
    opened = hip.g1_batch_add(mine[0], mine[1])
    xy, inf = hip.g1_batch_to_affine(opened)
    for i in range(n):
        x, y = limbs_to_ints(xy[8 * i:8 * i + 8])
        assert (pyref.from_mont(3, x), pyref.from_mont(3, y)) == pyref.g1_mul(pyref.G, vals[i])

    def checks(shares):
        out = []
        for p in (0, 1):
            o = np.zeros(12 * n, dtype=np.uint64); e.point_mac_check_shares(n, keys[p], opened, shares[p], o); out.append(o)
        ok = np.zeros(n, dtype=np.uint8); e.point_mac_verify(n, out[0], out[1], ok)
        return out, ok

    chk, ok = checks(sh)
    assert ok.tolist() == [1] * n
    # oracle agreement on the per-party check points
    for p in (0, 1):
        kv = oracle.g1_batch_scalar_mul(opened, np.tile(keys[p], n))
        macs = np.ascontiguousarray(sh[p].reshape(-1, 24)[:, 12:].reshape(-1))
        assert affine_equal(hip, oracle, chk[p], oracle.g1_batch_add(kv, oracle.g1_neg(macs)))
    bad = [sh[0].copy(), sh[1].copy()]
    bad[0][24 * 2 + 12:24 * 3] = jac([pyref.g1_mul(pyref.G, 999)], [1])
    _, ok = checks(bad)
    assert ok.tolist() == [1, 1, 0, 1, 1, 1]


def test_point_commitments_on_gpu(hip, oracle):
    """K9 vs the oracle's commit over to_bytes(point) || BE(blinder) (commitment.rs:71-86 with one value), incl. the identity."""
    n = 33
    pts, P = random_points(n, 21)
    bl = mont_array(0, rand_values(0, n, 22))
    out = np.zeros(4 * n, dtype=np.uint64)
    hip.eng(0).commit_points_sha3(n, P, bl, out)
    comp = oracle.g1_to_bytes(P).tobytes()
    for i in range(n):
        want = oracle.commit_bytes(0, comp[32 * i:32 * i + 32], bl[4 * i:4 * i + 4].copy())
        assert np.array_equal(out[4 * i:4 * i + 4], want), i
    # and against hashlib directly for one element
    import hashlib
    h = hashlib.sha3_256(pyref.g1_compress(pts[0]) + pyref.to_bytes_be(0, pyref.from_mont(0, limbs_to_ints(bl[:4])[0]))).digest()
    assert pyref.from_mont(0, limbs_to_ints(out[:4])[0]) == int.from_bytes(h, "big") % pyref.P[0]


def test_glv_scalar_mul_many_scalars(hip, oracle):
    """GLV decomposition robustness: 1500 random scalars plus values around lambda, r/2, r/3 and powers of two, against the
    oracle's plain double-and-add (affine comparison) -- and the two device algorithms agree with each other."""
    lam = 0xb3c4d79d41a917585bfc41088d8daaa78b17ea66b99c90dd
    r = pyref.RORD
    special = [0, 1, 2, 15, 16, r - 1, r - 2, lam, lam - 1, lam + 1, r - lam, r // 2, r // 2 + 1, r // 3, (1 << 127), (1 << 128) - 1, 1 << 128, (1 << 253)]
    ks = special + rand_values(0, 1500 - len(special), 77)
    n = len(ks)
    pts, P = random_points(n, 78, with_identity=True)
    S = mont_array(0, ks)
    got = hip.g1_batch_scalar_mul(P, S)
    want = oracle.g1_batch_scalar_mul(P, S)
    assert affine_equal(hip, oracle, got, want)


@pytest.mark.parametrize("n", [0, 1, 2, 255, 256, 257, 1500])
def test_point_sums(hip, oracle, n):
    """arkmpc_g1_sum / arkmpc_pointshare_sum vs the oracle's left fold (group law is associative: affine equality)."""
    pts, P = random_points(max(n, 1), 31 + n)
    P = np.ascontiguousarray(P[:12 * n])
    out = np.zeros(12, dtype=np.uint64)
    hip.eng(0).g1_sum(n, P if n else np.zeros(12, dtype=np.uint64), out)
    want = oracle.g1_sum(P) if n else oracle.g1_identity()
    assert affine_equal(hip, oracle, out, want)
    if n and n % 2 == 0:
        m = n // 2
        out2 = np.zeros(24, dtype=np.uint64)
        hip.eng(0).pointshare_sum(m, P, out2)
        assert affine_equal(hip, oracle, out2[:12].copy(), oracle.g1_sum(P, stride=24, off=0))
        assert affine_equal(hip, oracle, out2[12:].copy(), oracle.g1_sum(P, stride=24, off=12))


def test_from_bytes(hip, oracle):
    """arkmpc_g1_from_bytes vs the oracle: round trip of to_bytes, (x, y, 1) outputs bit-equal, invalid encodings flagged."""
    n = 200
    pts, P = random_points(n, 41)
    data = hip.g1_to_bytes(P)
    bad = [int(pyref.Q).to_bytes(32, "little"), bytes(31) + b"\xc0", (5).to_bytes(32, "little"), bytes(31) + b"\x40", (1).to_bytes(32, "little")]
    data = np.concatenate([data, np.frombuffer(b"".join(bad), dtype=np.uint8)])
    m = len(data) // 32
    out = np.zeros(12 * m, dtype=np.uint64); ok = np.zeros(m, dtype=np.uint8)
    hip.eng(0).g1_from_bytes(m, data, out, ok)
    want, want_ok = oracle.g1_from_bytes(data)
    assert np.array_equal(ok, want_ok) and np.array_equal(out, want)       # no addition chain involved: exact limbs
    assert ok[:n].all() and affine_equal(hip, oracle, out[:12 * n], P)
    assert ok[n:].tolist() == want_ok[n:].tolist()


def test_empty_batches_on_point_entry_points(hip, oracle):
    """n = 0 through the newer point entry points: no launch, no error, MSM of nothing is the identity."""
    eng = hip.eng(0)
    z12, z24, zb, zk = np.zeros(12, dtype=np.uint64), np.zeros(24, dtype=np.uint64), np.zeros(32, dtype=np.uint8), np.zeros(1, dtype=np.uint8)
    eng.g1_from_bytes(0, zb, z12, zk)
    out = np.zeros(24, dtype=np.uint64)
    eng.g1_msm_authenticated(0, z12, np.zeros(8, dtype=np.uint64), out)
    assert np.array_equal(out[:12], oracle.g1_identity()) and np.array_equal(out[12:], oracle.g1_identity())
    eng.g1_generator_mul(0, np.zeros(4, dtype=np.uint64), z12)
    eng.g1_to_bytes(0, z12, zb)
    cap = eng.wire_frame_bound(0); buf = np.zeros(cap, dtype=np.uint8)
    ln = eng.wire_encode_bytes32(1, 5, 0, zb, buf, cap)
    assert buf[8:ln].tobytes() == b'{"result_id":5,"payload":{"PointBatch":[]}}'


# ---- round 2: the hand-scheduled scalar-mul pipeline's exceptional lanes ----------------------------------------------------
def _asm_blinding_points():
    """R0 and C = 2^130 R0 of the window loop, read back from the GENERATED header (Montgomery limbs) -- the test does not trust the generator's Python"""
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    txt = open(os.path.join(root, "ark-mpc_amd", "csrc", "ec_asm_kernels.inc")).read()
    def const(name):
        m = re.search(r"G1_ASM_%s\[8\] = \{([^}]*)\}" % name, txt)
        limbs = [int(x.strip().rstrip("u"), 16) for x in m.group(1).split(",")]
        return pyref.from_mont(3, sum(l << (32 * i) for i, l in enumerate(limbs)))
    R0 = (const("R0X"), const("R0Y"))
    negC = (const("NCX"), const("NCY"))
    return R0, (negC[0], pyref.P[3] - negC[1])


def _exceptional_cases():
    R0, C = _asm_blinding_points()
    r = pyref.RORD
    neg = lambda P: (P[0], pyref.P[3] - P[1])
    R32 = pyref.g1_mul(R0, 32)
    k25 = (1 << 125) + 12345                                           # below the GLV lattice's reach: decomposes as (k, 0); window 25 digit = +1
    flagged = [(R32, k25), (neg(R32), k25), (R32, (1 << 125)), (neg(pyref.g1_add(C, C)), 1), (neg(C), 1), (C, r - 1)]
    plain = [(R0, 1), (R0, 0), (None, 7), (R32, r - k25)]            # the last one decomposes differently (k1 is not -k25): an ordinary lane
    return flagged, plain


@pytest.mark.parametrize("limbs", ["29", "32"])
def test_crafted_inputs_really_hit_the_exceptional_path(tmp_path, limbs):
    """Both forms of the window loop (nine 29-bit limbs: the default, flag = final Z is 0 mod q; eight 32-bit limbs, ARKMPC_EC_LIMBS=32: flag = H is 0
    in an addition) must flag exactly the same lanes.  With the recomputation switched off (ARKMPC_EC_ASM_NOFIX=1, a test hook) every crafted lane must come out WRONG and every
    ordinary lane right: the inputs do reach H = 0 inside the loop, and nothing else does."""
    import os
    import subprocess
    import sys
    import textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    prog = textwrap.dedent("""
        import importlib, sys, os
        sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
        import numpy as np, pyref, oracle_api
        from helpers import mont_array, limbs_to_ints, EngineAdapter
        import test_gpu_curve as T
        hip = EngineAdapter(importlib.import_module("ark-mpc_amd")); ora = oracle_api.load()
        flagged, plain = T._exceptional_cases()
        cases = flagged + plain
        P = T.jac([c[0] for c in cases], [5 + i for i in range(len(cases))]); S = mont_array(0, [c[1] for c in cases])
        got = hip.g1_batch_scalar_mul(P, S); want = ora.g1_batch_scalar_mul(P, S)
        gx, gi = hip.g1_batch_to_affine(got); wx, wi = ora.g1_batch_to_affine(want)
        print([int(np.array_equal(gx[8 * i:8 * i + 8], wx[8 * i:8 * i + 8]) and gi[i] == wi[i]) for i in range(len(cases))])
    """ % (root, root))
    env = dict(os.environ, ARKMPC_EC_ASM_NOFIX="1", ARKMPC_EC_LIMBS=limbs)
    r = subprocess.run([sys.executable, "-c", prog], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    flagged, plain = _exceptional_cases()
    assert eval(r.stdout.strip().splitlines()[-1]) == [0] * len(flagged) + [1] * len(plain)


def test_both_limb_forms_of_the_pipeline_agree_with_the_oracle():
    """The 32-bit-limb kernels stay selectable (ARKMPC_EC_LIMBS=32); a mixed batch -- crafted exceptional lanes, the identity, zero scalars,
    random lanes -- and a PointShare x Scalar batch (two lanes per table column) must equal the oracle in that form too, and the canonical affine
    outputs of the two forms must be the same bytes."""
    import os
    import subprocess
    import sys
    import textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    prog = textwrap.dedent("""
        import importlib, sys, os, hashlib
        sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
        import numpy as np, pyref, oracle_api
        from helpers import mont_array, rand_values, EngineAdapter
        import test_gpu_curve as T
        hip = EngineAdapter(importlib.import_module("ark-mpc_amd")); ora = oracle_api.load()
        flagged, plain = T._exceptional_cases()
        rnd_pts, _ = T.random_points(150, 99, with_identity=False)
        pts = [c[0] for c in flagged + plain] + rnd_pts; ks = [c[1] for c in flagged + plain] + rand_values(0, 150, 100)
        P = T.jac(pts, [11 + 3 * i for i in range(len(pts))]); S = mont_array(0, ks)
        got = hip.g1_batch_scalar_mul(P, S)
        ok1 = T.affine_equal(hip, ora, got, ora.g1_batch_scalar_mul(P, S))
        n = len(pts) // 2
        sc = mont_array(0, rand_values(0, n, 101))
        got2 = hip.pointshare_mul_public(P[:24 * n], sc); want2 = ora.pointshare_mul_public(P[:24 * n], sc)
        ok2 = T.affine_equal(hip, ora, got2, want2)
        xy, inf = hip.g1_batch_to_affine(got); xy2, inf2 = hip.g1_batch_to_affine(got2)
        print(int(ok1), int(ok2), hashlib.sha256(xy.tobytes() + inf.tobytes() + xy2.tobytes() + inf2.tobytes()).hexdigest())
    """ % (root, root))
    outs = {}
    for limbs in ("29", "32"):
        r = subprocess.run([sys.executable, "-c", prog], capture_output=True, text=True, env=dict(os.environ, ARKMPC_EC_LIMBS=limbs), timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[limbs] = r.stdout.strip().splitlines()[-1].split()
        assert outs[limbs][:2] == ["1", "1"], (limbs, outs[limbs])
    assert outs["29"][2] == outs["32"][2]


def test_asm_loop_exceptional_lanes_are_recomputed_exactly(hip, oracle):
    """Inputs crafted against the loop's PUBLIC blinding point so that a mixed addition inside it meets H = 0 (the accumulator equals, or is
    the negative of, the table entry it adds): P = +-32 R0 with a scalar whose window-25 digit is 1 hits step 2 (accumulator = 2^5 R0);
    P = -2 C with scalar 1 makes the final correction a doubling; P = -C with scalar 1 makes it P - P.  The loop only flags such lanes and the
    finish kernel recomputes them on the compiled path, so the results must still be the exact group-law answers -- mixed into a batch of
    ordinary lanes.  Also: the identity as input, scalars 0, 1, r - 1."""
    R0, C = _asm_blinding_points()
    r = pyref.RORD
    assert pyref.g1_mul(R0, pow(2, 130, r)) == C                       # the header's constants are consistent with each other
    flagged, plain = _exceptional_cases()
    cases = flagged + plain
    rnd_pts, _ = random_points(54, 4242, with_identity=False)
    rnd_k = rand_values(0, 54, 4243)
    pts = [c[0] for c in cases] + rnd_pts
    ks = [c[1] for c in cases] + rnd_k
    order = np.random.RandomState(7).permutation(len(pts))             # exceptional lanes scattered inside the wave
    pts, ks = [pts[i] for i in order], [ks[i] for i in order]
    P = jac(pts, [3 + 7 * i for i in range(len(pts))])
    S = mont_array(0, ks)
    got = hip.g1_batch_scalar_mul(P, S)
    assert affine_equal(hip, oracle, got, oracle.g1_batch_scalar_mul(P, S))
    xy, inf = hip.g1_batch_to_affine(got)
    for i, (pt, k) in enumerate(zip(pts, ks)):
        want = None if pt is None else pyref.g1_mul(pt, k % r)
        if want is None:
            assert inf[i] == 1
        else:
            x, y = limbs_to_ints(xy[8 * i:8 * i + 8])
            assert (pyref.from_mont(3, x), pyref.from_mont(3, y)) == want, (i, k)


def test_generator_mul_digit_edges(hip, oracle):
    """BN254 fixed-base chain (signed 12-bit digits, 22 windows, additions on the hand-scheduled body): 0, +-1, +-2048 (the recoding threshold),
    2049, all-ones windows, one digit per window position, chains of carries, r - 1 -- against Python's group law on the generator."""
    r = pyref.RORD
    ks = [0, 1, 2, r - 1, r - 2, r - 2048, r - 2049, (1 << 253) % r, ((1 << 254) - 1) % r]
    for w in range(22):
        for d in (1, 2047, 2048, 2049, 4095):
            ks.append((d << (12 * w)) % r)
            ks.append((r - (d << (12 * w))) % r)
    ks.append(sum(2048 << (12 * w) for w in range(21)))
    ks.append(sum(2049 << (12 * w) for w in range(21)))
    ks += rand_values(0, 40, 4343)
    n = len(ks)
    S = mont_array(0, ks)
    got = np.zeros(12 * n, dtype=np.uint64); hip.eng(0).g1_generator_mul(n, S, got)
    xy, inf = hip.g1_batch_to_affine(got)
    for i, k in enumerate(ks):
        want = pyref.g1_mul(pyref.G, k)
        if want is None:
            assert inf[i]
        else:
            x, y = limbs_to_ints(xy[8 * i:8 * i + 8])
            assert not inf[i] and (pyref.from_mont(3, x), pyref.from_mont(3, y)) == want, hex(k)


@pytest.mark.parametrize("party", [0, 1])
@pytest.mark.parametrize("host_mode", [True, False])
def test_point_beaver_finish_equals_the_references_four_terms(pkg, hip, oracle, party, host_mode):
    """arkmpc_point_beaver_finish (the point-side K3, regrouped to ([a] + d) eG + ([c] + d[b]) G) against the reference's literal terms
    deG + d[bG] + [a]eG + [c]G (authenticated_curve.rs:703-713) evaluated with the ORACLE's per-op functions, on affine coordinates, for
    random opened values, triple shares and MAC key; host-pointer and device-pointer contexts."""
    n = 21
    r = pyref.RORD
    key = mont_array(0, rand_values(0, 1, 600 + party))
    d = mont_array(0, [0, 1, r - 1] + rand_values(0, n - 3, 601))
    _, eG = random_points(n, 602, with_identity=True)
    ta, tb, tc = (mont_array(0, rand_values(0, 2 * n, 603 + k)) for k in range(3))           # n ScalarShares each (share, mac interleaved)
    # literal sequence on the oracle
    bG = oracle.scalarshare_mul_generator(tb)                                                  # [b]G            (:696)
    deG = oracle.g1_batch_scalar_mul(eG, d)                                                    # d * eG          (:705)
    d2 = np.repeat(d.reshape(n, 4), 2, axis=0).reshape(-1)
    dbG = oracle.pointshare_mul_public(bG, d)                                                  # d [bG]          (:706)
    aeG = oracle.scalarshare_mul_point(ta, eG)                                                 # [a] eG          (:707)
    cG = oracle.scalarshare_mul_generator(tc)                                                  # [c] G           (:708)
    want = oracle.pointshare_add(oracle.pointshare_add_public(party, key, dbG, deG), oracle.pointshare_add(aeG, cG))     # (:710-713)
    e = pkg.Engine(0, device=0, host_buffers=host_mode)
    out = np.zeros(24 * n, dtype=np.uint64)
    if host_mode:
        e.point_beaver_finish(n, party, key, d, eG, ta, tb, tc, out)
    else:
        import torch
        dev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).cuda()
        o = torch.zeros(24 * n, dtype=torch.int64, device="cuda")
        e.point_beaver_finish(n, party, key, dev(d), dev(eG), dev(ta), dev(tb), dev(tc), o)
        e.sync()                                              # the context has its own stream
        out = o.cpu().numpy().view(np.uint64)
    assert affine_equal(hip, oracle, out, want)
    with pytest.raises(pkg.ArkMpcError):
        e.point_beaver_finish(n, 2, key, d, eG, ta, tb, tc, out)
    e.close()
