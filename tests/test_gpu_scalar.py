"""GPU parity: every scalar-side C-ABI entry point vs the CPU oracle, bit-exact, on seeded inputs.
Runs through the C ABI (include/arkmpc.h) in both buffer modes (host staging and device pointers)."""
import numpy as np
import pytest

import pyref
from helpers import (mont_array, from_mont_array, mixed_values, rand_values, limbs_to_ints, ints_to_limbs,
                     authenticated_shares, interleave_shares)

pytestmark = pytest.mark.gpu

FIDS = [0, 1, 2, 3, 4]
NAMES = {0: "bn254_fr", 1: "bls12_381_fr", 2: "curve25519_fr", 3: "bn254_fq", 4: "curve25519_fq"}
SIZES = [1, 63, 257, 1000]


@pytest.fixture(scope="module")
def engines(pkg):
    return {fid: pkg.Engine(fid, device=0, host_buffers=True) for fid in FIDS}


def _z(n, w):
    return np.zeros(n * w, dtype=np.uint64)


@pytest.mark.parametrize("fid", FIDS)
@pytest.mark.parametrize("n", SIZES)
def test_scalar_ops(engines, oracle, fid, n):
    e = engines[fid]
    a = mont_array(fid, mixed_values(fid, n, seed=11 * n + fid))
    b = mont_array(fid, list(reversed(mixed_values(fid, n, seed=13 * n + fid))))
    for name, ora in [("scalar_add", oracle.scalar_add), ("scalar_sub", oracle.scalar_sub), ("scalar_mul", oracle.scalar_mul)]:
        out = _z(n, 4)
        getattr(e, name)(n, a, b, out)
        assert np.array_equal(out, ora(fid, a, b)), name
    out = _z(n, 4); e.scalar_neg(n, a, out)
    assert np.array_equal(out, oracle.scalar_neg(fid, a))
    out = _z(n, 4); e.open_combine(n, a, b, out)
    assert np.array_equal(out, oracle.open_combine(fid, a, b))
    out = _z(n, 4); e.scalar_to_canonical(n, a, out)
    assert np.array_equal(out, oracle.to_canonical(fid, a))
    raw = ints_to_limbs([(v * 3 + (1 << 255)) % (1 << 256) for v in limbs_to_ints(a)])  # includes values >= p
    out = _z(n, 4); e.scalar_from_canonical(n, raw, out)
    assert np.array_equal(out, oracle.from_canonical(fid, raw))
    outb = np.zeros(32 * n, dtype=np.uint8); e.scalar_to_bytes_be(n, a, outb)
    assert np.array_equal(outb, oracle.to_bytes_be(fid, a))


@pytest.mark.parametrize("fid", FIDS)
@pytest.mark.parametrize("n", [0, 1, 255, 257, 1000, 300001])
def test_scalar_and_share_reductions(engines, oracle, fid, n):
    """arkmpc_scalar_sum / _product / arkmpc_share_sum vs the oracle's left-to-right folds (scalar_result.rs:325-338, share.rs:103-111):
    empty input, one workgroup, two levels, more than RED_MAX_BLOCKS workgroups' worth of elements."""
    e = engines[fid]
    vals = mixed_values(fid, min(n, 2000), seed=17 * n + fid)
    a = mont_array(fid, vals)
    if n > 2000:                                              # large case: tile the small vector (python big-int setup stays cheap)
        a = np.ascontiguousarray(np.tile(a.reshape(-1, 4), ((n + 1999) // 2000, 1))[:n].reshape(-1))
    out = _z(1, 4); e.scalar_sum(n, a, out)
    assert np.array_equal(out, oracle.scalar_sum(fid, a))
    out = _z(1, 4); e.scalar_product(n, a, out)
    assert np.array_equal(out, oracle.scalar_product(fid, a))
    rec = np.ascontiguousarray(np.concatenate([a.reshape(-1, 4), a.reshape(-1, 4)[::-1]], axis=1).reshape(-1))
    out = _z(1, 8); e.share_sum(n, rec, out)
    assert np.array_equal(out, oracle.share_sum(fid, rec))


@pytest.mark.parametrize("fid", FIDS)
@pytest.mark.parametrize("n", [1, 300])
@pytest.mark.parametrize("party", [0, 1])
def test_share_ops(engines, oracle, fid, n, party):
    e = engines[fid]
    p = pyref.P[fid]
    key = mont_array(fid, [rand_values(fid, 1, 99)[0]])
    a = interleave_shares(mont_array(fid, mixed_values(fid, n, 1)), mont_array(fid, mixed_values(fid, n, 2)[::-1]))
    b = interleave_shares(mont_array(fid, rand_values(fid, n, 3)), mont_array(fid, mixed_values(fid, n, 4)))
    pub = mont_array(fid, mixed_values(fid, n, 5))
    out = _z(n, 8); e.share_add(n, a, b, out); assert np.array_equal(out, oracle.share_add(fid, a, b))
    out = _z(n, 8); e.share_sub(n, a, b, out); assert np.array_equal(out, oracle.share_sub(fid, a, b))
    out = _z(n, 8); e.share_neg(n, a, out); assert np.array_equal(out, oracle.share_neg(fid, a))
    out = _z(n, 8); e.share_mul_public(n, a, pub, out); assert np.array_equal(out, oracle.share_mul_public(fid, a, pub))
    out = _z(n, 8); e.share_add_public(n, party, key, a, pub, out)
    assert np.array_equal(out, oracle.share_add_public(fid, party, key, a, pub))
    out = _z(n, 8); e.share_sub_public(n, party, key, a, pub, out)
    assert np.array_equal(out, oracle.share_add_public(fid, party, key, a, pub, sub=True))
    out = _z(n, 4); e.share_extract(n, a, out)
    assert np.array_equal(out.reshape(-1, 4), a.reshape(-1, 8)[:, :4])


def _two_party_inputs(fid, n, seed):
    p = pyref.P[fid]
    key0, key1 = rand_values(fid, 2, seed + 1)
    key = (key0 + key1) % p
    x = mixed_values(fid, n, seed + 2)
    y = mixed_values(fid, n, seed + 3)[::-1]
    ta = rand_values(fid, n, seed + 4)
    tb = rand_values(fid, n, seed + 5)
    tc = [(u * v) % p for u, v in zip(ta, tb)]
    sh = {name: authenticated_shares(fid, vals, key, seed + 10 * i)
          for i, (name, vals) in enumerate([("x", x), ("y", y), ("a", ta), ("b", tb), ("c", tc)])}
    keys = [mont_array(fid, [key0]), mont_array(fid, [key1])]
    return x, y, key, keys, sh


@pytest.mark.parametrize("fid", FIDS)
@pytest.mark.parametrize("n", [1, 64, 777])
def test_beaver_mul_two_party(engines, oracle, fid, n):
    """authenticated_scalar.rs test_batch_mul (:1571-1594) restated: open(batch_mul(x, y)) == x*y, MACs verify,
    and every intermediate buffer equals the oracle's."""
    e = engines[fid]
    p = pyref.P[fid]
    x, y, key, keys, sh = _two_party_inputs(fid, n, seed=1000 + n)
    de, res, res_fused = [], [], []
    for party in (0, 1):
        out = _z(2 * n, 4)
        e.beaver_mask(n, sh["x"][party], sh["y"][party], sh["a"][party], sh["b"][party], out)
        assert np.array_equal(out, oracle.beaver_mask(fid, sh["x"][party], sh["y"][party], sh["a"][party], sh["b"][party]))
        de.append(out)
    opened = _z(2 * n, 4)
    e.open_combine(2 * n, de[0], de[1], opened)
    assert np.array_equal(opened, oracle.open_combine(fid, de[0], de[1]))
    d, ee = opened[:4 * n].copy(), opened[4 * n:].copy()
    for party in (0, 1):
        out = _z(n, 8)
        e.beaver_finish(n, party, keys[party], d, ee, sh["a"][party], sh["b"][party], sh["c"][party], out)
        want = oracle.beaver_finish(fid, party, keys[party], d, ee, sh["a"][party], sh["b"][party], sh["c"][party])
        assert np.array_equal(out, want)
        res.append(out)
        out2 = _z(n, 8)
        e.beaver_finish_fused(n, party, keys[party], de[party], de[1 - party], sh["a"][party], sh["b"][party], sh["c"][party], out2)
        assert np.array_equal(out2, want)
        # the reference's literal 9-pass sequence gives the same bits
        my_de9, out9 = oracle.batch_mul_9pass_local(fid, party, keys[party], sh["x"][party], sh["y"][party], sh["a"][party],
                                                    sh["b"][party], sh["c"][party], de[1 - party])
        assert np.array_equal(my_de9, de[party]) and np.array_equal(out9, want)
    # protocol algebra: shares sum to x*y, MAC shares sum to key*x*y
    r0 = np.asarray(res[0]).reshape(-1, 8); r1 = np.asarray(res[1]).reshape(-1, 8)
    prod = [(u + v) % p for u, v in zip(from_mont_array(fid, r0[:, :4].reshape(-1)), from_mont_array(fid, r1[:, :4].reshape(-1)))]
    macs = [(u + v) % p for u, v in zip(from_mont_array(fid, r0[:, 4:].reshape(-1)), from_mont_array(fid, r1[:, 4:].reshape(-1)))]
    assert prod == [(u * v) % p for u, v in zip(x, y)]
    assert macs == [(key * u * v) % p for u, v in zip(x, y)]


@pytest.mark.parametrize("fid", [0, 1, 2])
@pytest.mark.parametrize("n", [1, 500])
def test_open_authenticated_two_party(engines, oracle, fid, n):
    """authenticated_scalar.rs:278-354 restated for both parties, incl. commitment and the bad-MAC / bad-share cases
    (integration/src/authenticated_scalar.rs:49-75)."""
    e = engines[fid]
    p = pyref.P[fid]
    key0, key1 = rand_values(fid, 2, 31)
    key = (key0 + key1) % p
    keys = [mont_array(fid, [key0]), mont_array(fid, [key1])]
    vals = mixed_values(fid, n, 77)
    shares = list(authenticated_shares(fid, vals, key, 555))
    blinders = [mont_array(fid, [rand_values(fid, 1, 41)[0]]), mont_array(fid, [rand_values(fid, 1, 42)[0]])]

    def run(sh):
        mine = []
        for party in (0, 1):
            out = _z(n, 4); e.share_extract(n, sh[party], out); mine.append(out)
        opened, chk, comm = [], [], []
        for party in (0, 1):
            o = _z(n, 4); c = _z(n, 4)
            e.open_and_mac_check(n, keys[party], sh[party], mine[1 - party], o, c)
            o_ref = oracle.open_combine(fid, mine[party], mine[1 - party])
            assert np.array_equal(o, o_ref)
            assert np.array_equal(c, oracle.mac_check_shares(fid, keys[party], o_ref, sh[party]))
            c2 = _z(n, 4); e.mac_check_shares(n, keys[party], o, sh[party], c2)
            assert np.array_equal(c2, c)
            cm = e.commit_sha3(n, c, blinders[party])
            assert np.array_equal(cm, oracle.commit_scalars(fid, c, blinders[party]))
            opened.append(o); chk.append(c); comm.append(cm)
        oks = []
        for party in (0, 1):
            # verify the peer's commitment opens to the peer's values, then the sums (batch_verify_mac_check :201-220)
            recomputed = e.commit_sha3(n, chk[1 - party], blinders[1 - party])
            ok_comm = np.array_equal(recomputed, comm[1 - party])
            ok_sum = e.mac_verify(n, chk[party], chk[1 - party])
            assert ok_sum == oracle.mac_verify(fid, chk[party], chk[1 - party])
            oks.append(ok_comm and ok_sum)
        return opened, oks

    opened, oks = run(shares)
    assert oks == [True, True]
    assert from_mont_array(fid, opened[0]) == vals and np.array_equal(opened[0], opened[1])
    # corrupt one MAC (modify_mac, authenticated_scalar.rs:1090-1097) -> check must fail
    bad = [shares[0].copy(), shares[1].copy()]
    idx = n // 2
    bad[0][8 * idx + 4:8 * idx + 8] = mont_array(fid, [(pyref.from_mont(fid, limbs_to_ints(bad[0][8 * idx + 4:8 * idx + 8])[0]) + 1) % p])
    _, oks = run(bad)
    assert oks == [False, False]
    # corrupt one share (modify_share :1100-1110) -> opened value shifts, MAC check must fail
    bad = [shares[0].copy(), shares[1].copy()]
    bad[1][8 * idx:8 * idx + 4] = mont_array(fid, [12345])
    _, oks = run(bad)
    assert oks == [False, False]


def test_empty_and_errors(pkg, engines):
    e = engines[0]
    z = np.zeros(0, dtype=np.uint64)
    e.scalar_add(0, z, z, z)          # empty batches are no-ops (batch_mul returns vec![] :853-855)
    e.beaver_mask(0, z, z, z, z, z)
    assert e.mac_verify(0, z, z) is True
    with pytest.raises(pkg.ArkMpcError):
        e.share_add_public(4, 2, np.zeros(4, dtype=np.uint64), np.zeros(32, dtype=np.uint64), np.zeros(16, dtype=np.uint64), np.zeros(32, dtype=np.uint64))
    with pytest.raises(pkg.ArkMpcError):
        pkg.Engine(7)
    fq = engines[3]
    with pytest.raises(pkg.ArkMpcError):  # curve ops need the Fr context
        fq.g1_neg(1, np.zeros(12, dtype=np.uint64), np.zeros(12, dtype=np.uint64))


@pytest.mark.parametrize("fid", [0])
def test_device_pointer_mode_and_views(pkg, oracle, fid):
    """device-pointer mode on torch-owned memory + the share-view (split layout) entry points."""
    import torch
    n = 1000
    e = pkg.Engine(fid, device=0, host_buffers=False, stream=torch.cuda.current_stream().cuda_stream)
    x, y, key, keys, sh = _two_party_inputs(fid, n, seed=4242)
    dev = lambda a: torch.from_numpy(a.view(np.int64)).cuda()
    host = lambda t: t.cpu().numpy().view(np.uint64)
    for party in (0, 1):
        dx, dy, da, db, dc = (dev(sh[k][party]) for k in "xyabc")
        out = torch.zeros(2 * n * 4, dtype=torch.int64, device="cuda")
        e.beaver_mask(n, dx, dy, da, db, out)
        torch.cuda.synchronize()
        want_de = oracle.beaver_mask(fid, sh["x"][party], sh["y"][party], sh["a"][party], sh["b"][party])
        assert np.array_equal(host(out), want_de)
        # split layout: all shares then all macs
        def split(a):
            r = a.reshape(-1, 8)
            return dev(np.ascontiguousarray(r[:, :4]).reshape(-1)), dev(np.ascontiguousarray(r[:, 4:]).reshape(-1))
        xs, _ = split(sh["x"][party]); ys, _ = split(sh["y"][party])
        a_s, a_m = split(sh["a"][party]); b_s, b_m = split(sh["b"][party]); c_s, c_m = split(sh["c"][party])
        out_v = torch.zeros_like(out)
        e.beaver_mask_v(n, xs, 4, ys, 4, a_s, 4, b_s, 4, out_v)
        torch.cuda.synchronize()
        assert np.array_equal(host(out_v), want_de)
        peer = 1 - party
        peer_de = dev(oracle.beaver_mask(fid, sh["x"][peer], sh["y"][peer], sh["a"][peer], sh["b"][peer]))
        o_s = torch.zeros(n * 4, dtype=torch.int64, device="cuda"); o_m = torch.zeros_like(o_s)
        e.beaver_finish_fused_v(n, party, keys[party], out, peer_de, a_s, a_m, 4, b_s, b_m, 4, c_s, c_m, 4, o_s, o_m, 4)
        o_aos = torch.zeros(n * 8, dtype=torch.int64, device="cuda")
        e.beaver_finish_fused(n, party, keys[party], out, peer_de, da, db, dc, o_aos)
        torch.cuda.synchronize()
        opened = oracle.open_combine(fid, want_de, host(peer_de))
        want = oracle.beaver_finish(fid, party, keys[party], opened[:4 * n].copy(), opened[4 * n:].copy(),
                                    sh["a"][party], sh["b"][party], sh["c"][party])
        assert np.array_equal(host(o_aos), want)
        w = want.reshape(-1, 8)
        assert np.array_equal(host(o_s).reshape(-1, 4), w[:, :4]) and np.array_equal(host(o_m).reshape(-1, 4), w[:, 4:])
    # misaligned device pointer is rejected, not dereferenced
    t = torch.zeros(64, dtype=torch.int64, device="cuda")
    with pytest.raises(pkg.ArkMpcError):
        e.scalar_add(1, t.data_ptr() + 8, t.data_ptr() + 8, t.data_ptr() + 8)
    e.close()


@pytest.mark.parametrize("fid", [0, 1, 2, 3])
@pytest.mark.parametrize("layout", ["aos", "split"])
def test_hand_scheduled_finish_matches_cpp_kernel_at_scale(pkg, oracle, fid, layout):
    """The hand-scheduled K2+K3 body (asm_kernels.inc) vs the plain C++ kernel (unfused entry point, which never
    takes the asm path) on 2^18 + 77 random gates, every word compared; plus an oracle spot check. Hazard mistakes in
    hand-written gfx950 code show up as wrong words on SOME waves, so the comparison is exhaustive and repeated."""
    import torch
    n = (1 << 18) + 77
    e = pkg.Engine(fid, device=0, stream=torch.cuda.current_stream().cuda_stream)
    g = torch.Generator(device="cuda"); g.manual_seed(1234 + fid)
    def rnd(cnt):
        raw = torch.randint(-(2**63), 2**63 - 1, (4 * cnt,), dtype=torch.int64, device="cuda", generator=g)
        out = torch.empty_like(raw); e.scalar_from_canonical(cnt, raw, out); return out
    key = rnd(1).cpu().numpy().view(np.uint64).copy()
    my_de, peer_de = rnd(2 * n), rnd(2 * n)
    cols = {k: (rnd(n), rnd(n)) for k in "abc"}            # (share column, mac column)
    aos = {k: torch.cat([s.view(n, 4), m.view(n, 4)], dim=1).contiguous().view(-1) for k, (s, m) in cols.items()}
    opened = torch.empty_like(my_de); e.open_combine(2 * n, my_de, peer_de, opened)
    for party in (0, 1):
        ref = torch.empty(8 * n, dtype=torch.int64, device="cuda")
        e.beaver_finish(n, party, key, opened[:4 * n], opened[4 * n:], aos["a"], aos["b"], aos["c"], ref)      # C++ kernel
        for rep in range(3):
            if layout == "aos":
                out = torch.zeros(8 * n, dtype=torch.int64, device="cuda")
                e.beaver_finish_fused(n, party, key, my_de, peer_de, aos["a"], aos["b"], aos["c"], out)
                got = out
            else:
                o_s = torch.zeros(4 * n, dtype=torch.int64, device="cuda"); o_m = torch.zeros_like(o_s)
                e.beaver_finish_fused_v(n, party, key, my_de, peer_de, cols["a"][0], cols["a"][1], 4, cols["b"][0], cols["b"][1], 4,
                                        cols["c"][0], cols["c"][1], 4, o_s, o_m, 4)
                got = torch.cat([o_s.view(n, 4), o_m.view(n, 4)], dim=1).contiguous().view(-1)
            torch.cuda.synchronize()
            assert torch.equal(got, ref), "asm body differs from the C++ kernel (party %d, rep %d)" % (party, rep)
        # oracle spot check on the first 512 and last 77 gates
        h = lambda t: t.cpu().numpy().view(np.uint64).copy()
        for lo, hi in ((0, 512), (n - 77, n)):
            sl4 = lambda t: h(t[4 * lo:4 * hi]); sl8 = lambda t: h(t[8 * lo:8 * hi])
            want = oracle.beaver_finish(fid, party, key, sl4(opened[:4 * n]), sl4(opened[4 * n:]), sl8(aos["a"]), sl8(aos["b"]), sl8(aos["c"]))
            assert np.array_equal(sl8(ref), want)
    e.close()


@pytest.mark.parametrize("fid", FIDS)
@pytest.mark.parametrize("n", [1, 7, 8, 9, 1000])
def test_batch_inverse(engines, oracle, fid, n):
    a = mont_array(fid, mixed_values(fid, n, seed=3 * n + fid))      # zeros included
    out = _z(n, 4); engines[fid].scalar_batch_inverse(n, a, out)
    assert np.array_equal(out, oracle.scalar_batch_inverse(fid, a))
    prod = _z(n, 4); engines[fid].scalar_mul(n, a, out, prod)         # a * a^-1 == 1 (or 0)
    one = mont_array(fid, [1])
    for i in range(n):
        assert np.array_equal(prod[4 * i:4 * i + 4], one) or not a[4 * i:4 * i + 4].any()


def test_layout_converters(engines):
    n = 777
    e = engines[0]
    aos = np.arange(8 * n, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)
    s, m, back = _z(n, 4), _z(n, 4), _z(n, 8)
    e.share_split(n, aos, s, m)
    assert np.array_equal(s.reshape(-1, 4), aos.reshape(-1, 8)[:, :4]) and np.array_equal(m.reshape(-1, 4), aos.reshape(-1, 8)[:, 4:])
    e.share_join(n, s, m, back)
    assert np.array_equal(back, aos)


@pytest.mark.parametrize("fid", [0, 1])
@pytest.mark.parametrize("n", [1, 7, 2048, 2049, 70000])
def test_prefix_product_scan(engines, oracle, fid, n):
    """parallel inclusive scan vs the oracle's sequential chain (bit-exact: products of canonical residues are unique),
    across the one-block, two-level and ragged cases"""
    a = mont_array(fid, rand_values(fid, n, seed=5 * n + fid))
    out = _z(n, 4); engines[fid].scalar_prefix_product(n, a, out)
    assert np.array_equal(out, oracle.scalar_prefix_product(fid, a))


def test_thread_safety_same_and_separate_contexts(pkg, oracle):
    """include/arkmpc.h THREADING: calls on one context are serialised internally; different contexts are independent
    (gates may run on rayon workers concurrently, multi_threaded/executor.rs:208-217)."""
    import threading
    fid, n = 0, 5000
    shared = pkg.Engine(fid, device=0, host_buffers=True)
    own = [pkg.Engine(fid, device=0, host_buffers=True) for _ in range(2)]
    a = [mont_array(fid, rand_values(fid, n, 100 + t)) for t in range(2)]
    b = [mont_array(fid, rand_values(fid, n, 200 + t)) for t in range(2)]
    want = [oracle.scalar_mul(fid, a[t], b[t]) for t in range(2)]
    errors = []

    def work(t):
        try:
            for it in range(40):
                for eng in (shared, own[t]):
                    out = _z(n, 4)
                    eng.scalar_mul(n, a[t], b[t], out)
                    if not np.array_equal(out, want[t]):
                        errors.append((t, it))
        except Exception as ex:      # noqa: BLE001
            errors.append((t, repr(ex)))

    ths = [threading.Thread(target=work, args=(t,)) for t in range(2)]
    for th in ths: th.start()
    for th in ths: th.join()
    assert errors == []
    for e in [shared] + own:
        e.close()


def test_fill_records(pkg):
    """arkmpc_fill: n copies of a 4 / 8 / 12-word record (vec![value; n] of a preprocessing source)."""
    eng = pkg.Engine(0, device=0, host_buffers=True)
    for words in (2, 4, 8, 12):
        rec = np.arange(1, words + 1, dtype=np.uint64) * np.uint64(0x0123456789ABCDEF)
        for n in (0, 1, 5, 1000):
            out = np.zeros(max(n, 1) * words, dtype=np.uint64)
            eng.fill(n, rec, out)
            assert np.array_equal(out[:n * words], np.tile(rec, n))
    with pytest.raises(pkg.ArkMpcError):
        eng.fill(4, np.zeros(3, dtype=np.uint64), np.zeros(12, dtype=np.uint64))


def test_one_context_shared_by_many_threads(pkg, oracle):
    """a context is internally locked, so the gate closures of a multi-threaded executor
    (fabric/executor/multi_threaded) may share one.  8 threads x 40 calls on ONE host-buffer context (ctypes releases the GIL);
    every result must equal the oracle's."""
    import threading
    fid, n = 0, 3000
    eng = pkg.Engine(fid, device=0, host_buffers=True)
    errs = []
    def work(tid):
        try:
            a = mont_array(fid, rand_values(fid, n, 7000 + tid)); b = mont_array(fid, rand_values(fid, n, 7100 + tid))
            want_mul, want_add = oracle.scalar_mul(fid, a, b), oracle.scalar_add(fid, a, b)
            out = np.zeros(4 * n, dtype=np.uint64)
            for it in range(20):
                eng.scalar_mul(n, a, b, out)
                if not np.array_equal(out, want_mul): errs.append((tid, it, "mul"))
                eng.scalar_add(n, a, b, out)
                if not np.array_equal(out, want_add): errs.append((tid, it, "add"))
        except Exception as ex:                     # noqa: BLE001
            errs.append((tid, repr(ex)))
    ts = [threading.Thread(target=work, args=(t,)) for t in range(8)]
    for t in ts: t.start()
    for t in ts: t.join()
    assert not errs, errs[:3]


@pytest.mark.parametrize("host_mode", [True, False])
def test_mac_verify_async_is_sticky_until_collected(pkg, host_mode):
    """arkmpc_mac_verify_async / _result: several ranges verified with ONE synchronisation; a failure anywhere is reported by
    the next collection (and only that one), whichever range it was in and however many elements failed."""
    fid, n = 0, 3000
    e = pkg.Engine(fid, device=0, host_buffers=host_mode)
    mine = mont_array(fid, rand_values(fid, n, 901))
    peer = np.zeros_like(mine)
    eh = e if host_mode else pkg.Engine(fid, device=0, host_buffers=True)
    eh.scalar_neg(n, mine, peer)                                  # mine + peer == 0 everywhere
    if host_mode:
        A, B = mine, peer
    else:
        import torch
        A, B = torch.from_numpy(mine.view(np.int64)).cuda(), torch.from_numpy(peer.view(np.int64)).cuda()
    for _ in range(3):
        e.mac_verify_async(n, A, B)
    assert e.mac_verify_result() is True
    bad = peer.copy(); bad[4 * 1234] ^= np.uint64(1)
    Bb = bad if host_mode else torch.from_numpy(bad.view(np.int64)).cuda()
    e.mac_verify_async(n, A, B); e.mac_verify_async(n, A, Bb); e.mac_verify_async(n, A, B)
    assert e.mac_verify_result() is False                          # the middle range failed
    assert e.mac_verify_result() is True                           # collected: the flag is clear again
    all_bad = mont_array(fid, rand_values(fid, n, 902))
    Ba = all_bad if host_mode else torch.from_numpy(all_bad.view(np.int64)).cuda()
    assert e.mac_verify(n, A, Ba) is False                         # every element fails: one gate-admitted store
    assert e.mac_verify(n, A, Bb) is False                         # the gate re-opened after the failure was collected
    assert e.mac_verify(n, A, B) is True
    e.close()


@pytest.mark.parametrize("fid", [0, 1])
def test_open_and_mac_check_on_columns_equals_aos(pkg, oracle, fid):
    """arkmpc_open_and_mac_check_v on split columns and on an AoS view (stride 8) == the AoS entry point == the oracle."""
    import torch
    n = 1237
    e = pkg.Engine(fid, device=0, stream=torch.cuda.current_stream().cuda_stream)
    p = pyref.P[fid]
    key = rand_values(fid, 1, 77)[0]
    vals = mixed_values(fid, n, 78)
    sh0, sh1 = authenticated_shares(fid, vals, key, 79)
    kk = mont_array(fid, [rand_values(fid, 1, 80)[0]])
    peer = np.ascontiguousarray(sh1.reshape(n, 8)[:, :4]).reshape(-1)
    want_o = oracle.open_combine(fid, np.ascontiguousarray(sh0.reshape(n, 8)[:, :4]).reshape(-1), peer)
    want_c = oracle.mac_check_shares(fid, kk, want_o, sh0)
    dev = lambda a: torch.from_numpy(a.view(np.int64).copy()).cuda()
    d_aos, d_peer = dev(sh0), dev(peer)
    sc, mc = torch.empty(4 * n, dtype=torch.int64, device="cuda"), torch.empty(4 * n, dtype=torch.int64, device="cuda")
    e.share_split(n, d_aos, sc, mc)
    for share_col, mac_col, stride in ((sc, mc, 4), (d_aos, d_aos[4:], 8)):
        o = torch.zeros(4 * n, dtype=torch.int64, device="cuda"); c = torch.zeros_like(o)
        e.open_and_mac_check_v(n, kk, share_col, mac_col, stride, d_peer, o, c)
        torch.cuda.synchronize()
        assert np.array_equal(o.cpu().numpy().view(np.uint64), want_o) and np.array_equal(c.cpu().numpy().view(np.uint64), want_c)
    with pytest.raises(pkg.ArkMpcError):
        e.open_and_mac_check_v(n, kk, sc, mc, 3, d_peer, o, c)
    e.close()


def test_event_orders_one_contexts_stream_after_anothers(pkg):
    """arkmpc_event_record / _wait: context B (own stream) consumes what context A (own stream) is still computing, with no host
    synchronisation in between -- the shape of the mirror's device link (one party's masked values feeding the peer's K3)."""
    import torch
    fid, n = 0, 1 << 18
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    A = pkg.Engine(fid, device=0, stream=sa.cuda_stream)
    B = pkg.Engine(fid, device=0, stream=sb.cuda_stream)
    x = torch.from_numpy(mont_array(fid, rand_values(fid, n, 77)).view(np.int64)).cuda()
    inv = torch.zeros_like(x); prod = torch.zeros_like(x)
    torch.cuda.synchronize()
    for _ in range(8):                                             # a few ms of work queued on A's stream
        A.scalar_batch_inverse(n, x, inv)
    ev = A.event_record()
    B.event_wait(ev)                                               # returns at once; B's stream waits on the device
    B.scalar_mul(n, x, inv, prod)
    B.sync()
    A.event_destroy(ev)
    one = mont_array(fid, [1])
    got = prod.cpu().numpy().view(np.uint64).reshape(n, 4)
    assert (got == one.reshape(1, 4)).all()
    with pytest.raises(pkg.ArkMpcError):
        B.event_wait(None)
    A.close(); B.close()


def test_error_text_is_safe_under_concurrent_failures(pkg):
    """arkmpc_last_error returns a per-thread copy taken under the error-text lock: threads that keep failing (argument errors from the
    batch carrier, which reports without the context lock, and from a guarded entry point) while others read the text must never crash
    or see a torn string."""
    import ctypes
    import threading
    eng = pkg.Engine(0, device=0)
    lib, h = eng.lib, eng.h
    seen, errs = set(), []
    stop = threading.Event()

    def fail_carrier():
        out = ctypes.c_void_p()
        while not stop.is_set():
            if lib.arkmpc_batch_create(h, 99, 0, ctypes.c_size_t(4), ctypes.byref(out)) == 0: errs.append("carrier accepted kind 99")

    def fail_guarded():
        while not stop.is_set():
            if lib.arkmpc_kernel_timer_arm(h, ctypes.c_int(-5)) == 0: errs.append("timer accepted slot -5")

    def read():
        for _ in range(20000):
            msg = lib.arkmpc_last_error(h)
            seen.add(msg.decode() if msg else "")

    ts = [threading.Thread(target=f) for f in (fail_carrier, fail_guarded, read, read)]
    for t in ts: t.start()
    ts[2].join(); ts[3].join(); stop.set(); ts[0].join(); ts[1].join()
    assert not errs
    assert seen <= {"", "element kind not defined for this context", "timer slot out of range"}, seen
    eng.close()


@pytest.mark.parametrize("fid", [0, 1])
def test_range_forms_dup_broadcast_and_column_ops(pkg, oracle, fid):
    """Round-3 pointer-level entry points vs the oracle: arkmpc_beaver_mask_to / _finish_fused_from on a RANGE of a larger batch (d and e addressed
    separately, written into the full d||e buffer), arkmpc_beaver_mask_dup (payload written twice), element stride 0 (a constant triple as ONE
    record), and the column forms of add_public / sub_public / mul_public on split columns AND on an AoS view (stride 8)."""
    import torch
    n, lo, cnt = 1000, 137, 611
    e = pkg.Engine(fid, device=0, host_buffers=False, stream=torch.cuda.current_stream().cuda_stream)
    x, y, key, keys, sh = _two_party_inputs(fid, n, seed=777 + fid)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).cuda()
    host = lambda t: t.cpu().numpy().view(np.uint64)
    cols = lambda a: (dev(a.reshape(-1, 8)[:, :4].reshape(-1)), dev(a.reshape(-1, 8)[:, 4:].reshape(-1)))
    party = 1
    S = {k: cols(sh[k][party]) for k in "xyabc"}
    want_de = oracle.beaver_mask(fid, sh["x"][party], sh["y"][party], sh["a"][party], sh["b"][party])
    off = 4 * 8 * lo                                                    # byte offset of element lo in a 32-byte column
    # range form of K1: gates [lo, lo + cnt) written into their place of the FULL d||e buffer
    full = torch.zeros(8 * n, dtype=torch.int64, device="cuda")
    p = lambda t, extra=0: t.data_ptr() + extra
    e.beaver_mask_to(cnt, p(S["x"][0], off), 4, p(S["y"][0], off), 4, p(S["a"][0], off), 4, p(S["b"][0], off), 4, p(full, off), p(full, 32 * n + off))
    torch.cuda.synchronize()
    got = host(full).reshape(2, n, 4)
    w = want_de.reshape(2, n, 4)
    assert np.array_equal(got[:, lo:lo + cnt], w[:, lo:lo + cnt]) and not got[:, :lo].any() and not got[:, lo + cnt:].any()
    # dup: the same payload twice
    d1 = torch.zeros(8 * n, dtype=torch.int64, device="cuda"); d2 = torch.zeros_like(d1)
    e.beaver_mask_dup(n, S["x"][0], 4, S["y"][0], 4, S["a"][0], 4, S["b"][0], 4, d1, d2)
    torch.cuda.synchronize()
    assert np.array_equal(host(d1), want_de) and np.array_equal(host(d2), want_de)
    # range form of K2+K3 reading the two parties' d / e slices in place
    peer_de = oracle.beaver_mask(fid, sh["x"][0], sh["y"][0], sh["a"][0], sh["b"][0])
    dpeer = dev(peer_de)
    o_s = torch.zeros(4 * n, dtype=torch.int64, device="cuda"); o_m = torch.zeros_like(o_s)
    e.beaver_finish_fused_from(cnt, party, keys[party], p(d1, off), p(d1, 32 * n + off), p(dpeer, off), p(dpeer, 32 * n + off),
                               p(S["a"][0], off), p(S["a"][1], off), 4, p(S["b"][0], off), p(S["b"][1], off), 4, p(S["c"][0], off), p(S["c"][1], off), 4,
                               p(o_s, off), p(o_m, off), 4)
    torch.cuda.synchronize()
    opened = oracle.open_combine(fid, want_de, peer_de)
    want = oracle.beaver_finish(fid, party, keys[party], opened[:4 * n].copy(), opened[4 * n:].copy(), sh["a"][party], sh["b"][party], sh["c"][party]).reshape(-1, 8)
    gs, gm = host(o_s).reshape(-1, 4), host(o_m).reshape(-1, 4)
    assert np.array_equal(gs[lo:lo + cnt], want[lo:lo + cnt, :4]) and np.array_equal(gm[lo:lo + cnt], want[lo:lo + cnt, 4:])
    assert not gs[:lo].any() and not gm[lo + cnt:].any()
    # element stride 0: the triple is ONE record broadcast to every gate (vec![share; n] of the dummy source)
    ta, tb, tc = (np.tile(sh[k][party][:8], n) for k in "abc")
    A1, B1, C1 = (cols(sh[k][party][:8]) for k in "abc")
    dz = torch.zeros(8 * n, dtype=torch.int64, device="cuda")
    e.beaver_mask_v(n, S["x"][0], 4, S["y"][0], 4, A1[0], 0, B1[0], 0, dz)
    want_dz = oracle.beaver_mask(fid, sh["x"][party], sh["y"][party], ta, tb)
    oz_s = torch.zeros(4 * n, dtype=torch.int64, device="cuda"); oz_m = torch.zeros_like(oz_s)
    e.beaver_finish_fused_v(n, party, keys[party], dz, dpeer, A1[0], A1[1], 0, B1[0], B1[1], 0, C1[0], C1[1], 0, oz_s, oz_m, 4)
    torch.cuda.synchronize()
    assert np.array_equal(host(dz), want_dz)
    op2 = oracle.open_combine(fid, want_dz, peer_de)
    wz = oracle.beaver_finish(fid, party, keys[party], op2[:4 * n].copy(), op2[4 * n:].copy(), ta, tb, tc).reshape(-1, 8)
    assert np.array_equal(host(oz_s).reshape(-1, 4), wz[:, :4]) and np.array_equal(host(oz_m).reshape(-1, 4), wz[:, 4:])
    with pytest.raises(pkg.ArkMpcError):                                  # an OUTPUT cannot be broadcast
        e.beaver_finish_fused_v(n, party, keys[party], dz, dpeer, A1[0], A1[1], 0, B1[0], B1[1], 0, C1[0], C1[1], 0, oz_s, oz_m, 0)
    # column forms of the public-operand ops: split columns and an AoS view of the same records
    pub = mont_array(fid, mixed_values(fid, n, 5))
    dpub = dev(pub)
    aos = dev(sh["a"][party])
    for name, ora in (("add", lambda: oracle.share_add_public(fid, party, keys[party], sh["a"][party], pub)),
                      ("sub", lambda: oracle.share_add_public(fid, party, keys[party], sh["a"][party], pub, sub=True)),
                      ("mul", lambda: oracle.share_mul_public(fid, sh["a"][party], pub))):
        wv = ora().reshape(-1, 8)
        for view in ("split", "aos"):
            rs = torch.zeros(4 * n, dtype=torch.int64, device="cuda"); rm = torch.zeros_like(rs)
            a_s, a_m, ast = (S["a"][0], S["a"][1], 4) if view == "split" else (aos, aos.data_ptr() + 32, 8)
            if name == "mul":
                e.share_mul_public_v(n, a_s, a_m, ast, dpub, rs, rm, 4)
            else:
                getattr(e, "share_%s_public_v" % name)(n, party, keys[party], a_s, a_m, ast, dpub, rs, rm, 4)
            torch.cuda.synchronize()
            assert np.array_equal(host(rs).reshape(-1, 4), wv[:, :4]) and np.array_equal(host(rm).reshape(-1, 4), wv[:, 4:]), (name, view)
    e.close()
