"""GPU parity for the variable-base MSM (arkmpc_g1_msm / arkmpc_g1_msm_authenticated): the bucket-method kernels through
the C ABI vs (a) the oracle's plain sum of scalar multiples and (b) a size-independent closed form: with bases k_i * G
the result must be (sum_i s_i * k_i mod r) * G, evaluated in Python integers.  Reference: CurvePoint::msm
(curve.rs:549-560), CurvePointResult::msm_authenticated (curve.rs:618-642)."""
import os

import numpy as np
import pytest

import pyref
from helpers import mont_array, rand_values, limbs_to_ints, EngineAdapter
from test_gpu_curve import jac, random_points, affine_equal

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip(pkg):
    return EngineAdapter(pkg)


def msm(hip, P, S):
    out = np.zeros(12, dtype=np.uint64)
    n = len(P) // 12
    hip.eng(0).g1_msm(n, P if n else np.zeros(12, dtype=np.uint64), S if n else np.zeros(4, dtype=np.uint64), out)
    return out


def affine_ints(hip, pt):
    xy, inf = hip.g1_batch_to_affine(np.ascontiguousarray(pt))
    if inf[0]:
        return None
    x, y = limbs_to_ints(xy[:8])
    return pyref.from_mont(3, x), pyref.from_mont(3, y)


@pytest.mark.parametrize("n", [0, 1, 2, 7, 33, 200])
def test_msm_small_vs_oracle(hip, oracle, n):
    pts, P = random_points(max(n, 1), 900 + n)
    P = np.ascontiguousarray(P[:12 * n])
    ks = ([0, 1, pyref.RORD - 1, 2, (1 << 253)] + rand_values(0, max(n, 5), 901 + n))[:n]
    S = mont_array(0, ks)
    want = oracle.g1_msm(P, S) if n else oracle.g1_identity()
    assert affine_equal(hip, oracle, msm(hip, P, S), want)


def test_msm_exceptional_bucket_members(hip, oracle):
    """The same base repeated with equal scalars lands twice in one bucket (doubling inside the mixed addition); a base and
    its negative with equal scalars cancel inside a bucket; identity bases and zero scalars are skipped."""
    n = 16
    pts, P = random_points(n, 950, with_identity=True)
    ks = rand_values(0, n, 951)
    P[12 * 3:12 * 4] = P[12 * 2:12 * 3]; ks[3] = ks[2]
    P[12 * 5:12 * 6] = oracle.g1_neg(P[12 * 4:12 * 5].copy()); ks[5] = ks[4]
    P[12 * 7:12 * 8] = jac([pts[6]], [12345]); ks[7] = ks[6]            # same point, different Jacobian representative
    ks[8] = 0
    S = mont_array(0, ks)
    assert affine_equal(hip, oracle, msm(hip, P, S), oracle.g1_msm(P, S))
    # everything cancels -> identity in arkworks' canonical form (1, 1, 0)
    Q = np.concatenate([P[:24], oracle.g1_neg(P[:24].copy())])
    SQ = mont_array(0, [ks[0], ks[1], ks[0], ks[1]])
    assert np.array_equal(msm(hip, Q, SQ), oracle.g1_identity())


def test_msm_asm_accumulation_flags_exceptional_members():
    """The hand-scheduled bucket accumulation (k_msm_accumulate_asm) only FLAGS a task whose chain of mixed additions meets H = 0; the compiled
    kernel then redoes it.  With that recomputation switched off (ARKMPC_MSM_ASM_NOFIX=1, a test hook) the repeated-base input must come out
    wrong and an ordinary input right -- i.e. the exceptional test above does travel through the flag, and nothing else does."""
    import os
    import subprocess
    import sys
    import textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    prog = textwrap.dedent("""
        import importlib, sys, os
        sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
        import numpy as np, oracle_api
        from helpers import mont_array, rand_values, EngineAdapter
        import test_gpu_msm as T
        hip = EngineAdapter(importlib.import_module("ark-mpc_amd")); ora = oracle_api.load()
        n = 16
        pts, P = T.random_points(n, 950, with_identity=False)
        ks = rand_values(0, n, 951)
        S = mont_array(0, ks)
        plain = T.affine_equal(hip, ora, T.msm(hip, P, S), ora.g1_msm(P, S))
        P[12 * 3:12 * 4] = P[12 * 2:12 * 3]; ks[3] = ks[2]
        S = mont_array(0, ks)
        repeated = T.affine_equal(hip, ora, T.msm(hip, P, S), ora.g1_msm(P, S))
        print([int(plain), int(repeated)])
    """ % (root, root))
    out = {}
    for nofix in ("1", "0"):
        r = subprocess.run([sys.executable, "-c", prog], capture_output=True, text=True, env=dict(os.environ, ARKMPC_MSM_ASM_NOFIX=nofix), timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        out[nofix] = eval(r.stdout.strip().splitlines()[-1])
    assert out["1"] == [1, 0] and out["0"] == [1, 1]


@pytest.mark.parametrize("c", [2, 4, 7, 11, 13, 16])
def test_msm_every_window_width(hip, oracle, c):
    """Force the window width: W = 254 // c + 1 signed digits, top-window carry included."""
    n = 60
    pts, P = random_points(n, 960)
    ks = [pyref.RORD - 1, (1 << 253) + (1 << 252), (1 << 254) % pyref.RORD, 1, 0] + rand_values(0, n - 5, 961 + c)
    S = mont_array(0, ks)
    os.environ["ARKMPC_MSM_C"] = str(c)
    try:
        got = msm(hip, P, S)
    finally:
        del os.environ["ARKMPC_MSM_C"]
    assert affine_equal(hip, oracle, got, oracle.g1_msm(P, S))


def device_bases(hip, ks):
    n = len(ks)
    P = np.zeros(12 * n, dtype=np.uint64)
    hip.eng(0).g1_generator_mul(n, mont_array(0, ks), P)
    return P


@pytest.mark.parametrize("n", [1000, 1 << 14])
def test_msm_closed_form_and_authenticated(hip, oracle, n):
    r = pyref.RORD
    ks = rand_values(0, n, 970)
    ss = rand_values(0, n, 971)
    ms = rand_values(0, n, 972)
    P = device_bases(hip, ks)
    got = msm(hip, P, mont_array(0, ss))
    assert affine_ints(hip, got) == pyref.g1_mul(pyref.G, sum(s * k for s, k in zip(ss, ks)) % r)
    shares = mont_array(0, [v for pair in zip(ss, ms) for v in pair])
    out = np.zeros(24, dtype=np.uint64)
    hip.eng(0).g1_msm_authenticated(n, P, shares, out)
    assert affine_ints(hip, out[:12]) == pyref.g1_mul(pyref.G, sum(s * k for s, k in zip(ss, ks)) % r)
    assert affine_ints(hip, out[12:]) == pyref.g1_mul(pyref.G, sum(m * k for m, k in zip(ms, ks)) % r)
    assert np.array_equal(out[:12], got)                                  # same order of additions -> same representative
    if n <= 1000:
        assert affine_equal(hip, oracle, out[:12].copy(), oracle.g1_msm(P, shares, stride=8, off=0))
        assert affine_equal(hip, oracle, out[12:].copy(), oracle.g1_msm(P, shares, stride=8, off=4))


def test_msm_skewed_inputs(hip):
    """One scalar for every point (a single bucket per window takes all n members) and one point for every scalar."""
    r = pyref.RORD
    n = 3000
    ks = rand_values(0, n, 980)
    s = rand_values(0, 1, 981)[0]
    P = device_bases(hip, ks)
    assert affine_ints(hip, msm(hip, P, mont_array(0, [s] * n))) == pyref.g1_mul(pyref.G, s * sum(ks) % r)
    ss = rand_values(0, n, 982)
    P1 = device_bases(hip, [ks[0]] * n)
    assert affine_ints(hip, msm(hip, P1, mont_array(0, ss))) == pyref.g1_mul(pyref.G, ks[0] * sum(ss) % r)
    bits = [i & 1 for i in range(n)]                                       # 0/1 scalars (bit vectors)
    assert affine_ints(hip, msm(hip, P, mont_array(0, bits))) == pyref.g1_mul(pyref.G, sum(k for k, b in zip(ks, bits) if b) % r)


@pytest.mark.parametrize("k", [1, 3, 64])
def test_msm_task_splitting(hip, oracle, k):
    """Force tiny tasks so that every bucket run is cut into many partial sums and goes through the combine tree."""
    n = 300
    pts, P = random_points(n, 990)
    ks = rand_values(0, n, 991)
    ks[10:200] = [ks[10]] * 190                                            # one long run in every window
    S = mont_array(0, ks)
    os.environ["ARKMPC_MSM_K"] = str(k); os.environ["ARKMPC_MSM_C"] = "5"
    try:
        got = msm(hip, P, S)
    finally:
        del os.environ["ARKMPC_MSM_K"]; del os.environ["ARKMPC_MSM_C"]
    assert affine_equal(hip, oracle, got, oracle.g1_msm(P, S))


def test_msm_random_configurations(hip):
    """60 random (n, window width, task size, group size, scalar distribution) combinations against the closed form
    (sum s_i k_i) G: run boundaries at multiples of the task size, empty buckets, single-window widths, skew."""
    import random
    rng = random.Random(4242)
    r = pyref.RORD
    eng = hip.eng(0)
    nmax = 1500
    ks = rand_values(0, nmax, 1001)
    P_all = device_bases(hip, ks)
    for trial in range(60):
        n = rng.choice([1, 2, 3, 17, 255, 256, 257, 511, 1024, rng.randrange(1, nmax)])
        c = rng.choice([2, 3, 5, 8, 9, 12, 15, 16])
        K = rng.choice([1, 2, 7, 16, 256])
        L = rng.choice([1, 2, 4, 8])
        dist = rng.randrange(4)
        if dist == 0: ss = [rng.randrange(r) for _ in range(n)]
        elif dist == 1: ss = [rng.choice([0, 1, 2, r - 1]) for _ in range(n)]
        elif dist == 2: ss = [rng.randrange(1 << rng.choice([1, 8, 64, 128, 200]))  for _ in range(n)]
        else:
            v = rng.randrange(r); ss = [v] * n
        os.environ["ARKMPC_MSM_C"] = str(c); os.environ["ARKMPC_MSM_K"] = str(K)
        if (1 << (c - 1)) % L == 0: os.environ["ARKMPC_MSM_L"] = str(L)
        try:
            got = msm(hip, np.ascontiguousarray(P_all[:12 * n]), mont_array(0, ss))
        finally:
            for k_ in ("ARKMPC_MSM_C", "ARKMPC_MSM_K", "ARKMPC_MSM_L"): os.environ.pop(k_, None)
        want = pyref.g1_mul(pyref.G, sum(s * k for s, k in zip(ss, ks)) % r)
        assert affine_ints(hip, got) == want, (trial, n, c, K, L, dist)


def test_msm_normalised_bases_fast_path(hip, oracle):
    """Bases given as (x, y, 1) / identity skip the batch inversion; a mix with one non-normalised point in the wave does not."""
    n = 300
    pts, _ = random_points(n, 1100, with_identity=True)
    ks = rand_values(0, n, 1101)
    S = mont_array(0, ks)
    Pn = jac(pts, [1] * n)                                                 # every z is 1 (identity stays (1,1,0))
    want = oracle.g1_msm(Pn, S)
    assert affine_equal(hip, oracle, msm(hip, Pn, S), want)
    zs = [1] * n; zs[137] = 9
    assert affine_equal(hip, oracle, msm(hip, jac(pts, zs), S), want)
