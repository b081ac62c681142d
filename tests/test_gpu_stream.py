"""GPU parity of the streaming host-to-host path (include/arkmpc.h arkmpc_hostmul_*, csrc/arkmpc_stream.inc): host arkworks records in,
host records out, every word against the CPU oracle (authenticated_scalar.rs:848-879).  The pipeline itself computes nothing new -- these
tests pin the plumbing: chunk boundaries (ragged last chunk), unaligned Rust-Vec-like pointers, aliased operands, pinned / pageable /
pre-registered buffers, the d||e progress counter, error paths that must end the session."""
import ctypes
import threading

import numpy as np
import pytest

import pyref
from helpers import FreshVA, mont_array, from_mont_array, mixed_values, rand_values, authenticated_shares

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _inputs(fid, n, seed, tile_from=None):
    """two-party authenticated shares of x, y and a Beaver triple.  Large n: a small python-int workload tiled (the oracle is the checker, so
    repeated gates cost nothing in coverage of the plumbing: positions still differ chunk by chunk through the rolled tiling)."""
    p = pyref.P[fid]
    m = n if tile_from is None else min(n, tile_from)
    key0, key1 = rand_values(fid, 2, seed + 1)
    key = (key0 + key1) % p
    x = mixed_values(fid, m, seed + 2)
    y = mixed_values(fid, m, seed + 3)[::-1]
    ta, tb = rand_values(fid, m, seed + 4), rand_values(fid, m, seed + 5)
    tc = [(u * v) % p for u, v in zip(ta, tb)]
    sh = {}
    for i, (name, vals) in enumerate([("x", x), ("y", y), ("a", ta), ("b", tb), ("c", tc)]):
        s0, s1 = authenticated_shares(fid, vals, key, seed + 10 * i)
        if m < n:
            reps = -(-n // m)
            s0 = np.ascontiguousarray(np.tile(s0.reshape(m, 8), (reps, 1))[:n].reshape(-1))
            s1 = np.ascontiguousarray(np.tile(s1.reshape(m, 8), (reps, 1))[:n].reshape(-1))
        sh[name] = (s0, s1)
    keys = [mont_array(fid, [key0]), mont_array(fid, [key1])]
    vals = (x, y, key)
    return vals, keys, sh


def _oracle_two_party(oracle, fid, n, keys, sh):
    big = n >= (1 << 16)
    mask = oracle.beaver_mask_mt if big else oracle.beaver_mask
    ode = [mask(fid, sh["x"][p], sh["y"][p], sh["a"][p], sh["b"][p]) for p in (0, 1)]
    want = []
    for p in (0, 1):
        if big:
            my_de, w = oracle.batch_mul_9pass_mt(fid, p, keys[p], sh["x"][p], sh["y"][p], sh["a"][p], sh["b"][p], sh["c"][p], ode[1 - p])
        else:
            my_de, w = oracle.batch_mul_9pass_local(fid, p, keys[p], sh["x"][p], sh["y"][p], sh["a"][p], sh["b"][p], sh["c"][p], ode[1 - p])
        assert np.array_equal(my_de, ode[p])
        want.append(w)
    return ode, want


def _run_two_party(eng, n, keys, sh, poll=False):
    """both parties through the session API on one context, the mock link = handing over the host d||e buffers"""
    de = [np.zeros(8 * n, dtype=np.uint64) for _ in (0, 1)]
    out = [np.zeros(8 * n, dtype=np.uint64) for _ in (0, 1)]
    seen = []
    ses = []                    # both sessions are open at once on the one context: they are independent objects, only the streams are shared
    for p in (0, 1):
        s = eng.hostmul_begin(n, sh["x"][p], sh["y"][p], sh["a"][p], sh["b"][p], sh["c"][p], de[p])
        ses.append(s)
    for p in (0, 1):
        if poll:
            g = eng.hostmul_poll_de(ses[p]); seen.append(g)
            assert 0 <= g <= n
        eng.hostmul_wait_de(ses[p])
        assert eng.hostmul_poll_de(ses[p]) == n
    for p in (0, 1):
        eng.hostmul_finish(ses[p], p, keys[p], de[1 - p], out[p])
    return de, out


@pytest.fixture(scope="module")
def engs(pkg):
    d = {fid: pkg.Engine(fid, device=0) for fid in (0, 1, 2)}
    _TRACKED.extend(d.values())
    return d


@pytest.mark.parametrize("fid", [0, 1, 2])
@pytest.mark.parametrize("n", [1, 2, 777, 16384, 70001])
def test_hostmul_bitexact_vs_oracle(engs, oracle, fid, n):
    """every word of d||e and of the result records, both parties; n = 70001 runs five chunks with a ragged last one"""
    e = engs[fid]
    p = pyref.P[fid]
    vals, keys, sh = _inputs(fid, n, seed=4000 + n, tile_from=3000)
    de, out = _run_two_party(e, n, keys, sh, poll=True)
    ode, want = _oracle_two_party(oracle, fid, n, keys, sh)
    for party in (0, 1):
        assert np.array_equal(de[party], ode[party]), "d||e of party %d" % party
        assert np.array_equal(out[party], want[party]), "result of party %d" % party
    m = min(n, 3000)
    x, y, key = vals
    r0, r1 = out[0].reshape(-1, 8)[:m], out[1].reshape(-1, 8)[:m]
    prod = [(u + v) % p for u, v in zip(from_mont_array(fid, r0[:, :4].reshape(-1)), from_mont_array(fid, r1[:, :4].reshape(-1)))]
    macs = [(u + v) % p for u, v in zip(from_mont_array(fid, r0[:, 4:].reshape(-1)), from_mont_array(fid, r1[:, 4:].reshape(-1)))]
    assert prod == [(u * v) % p for u, v in zip(x[:m], y[:m])]            # authenticated_scalar.rs test_batch_mul :1571-1594
    assert macs == [(key * u * v) % p for u, v in zip(x[:m], y[:m])]


@pytest.mark.parametrize("stream_kind", ["own", "torch_default"])
def test_hostmul_consecutive_sessions_on_recycled_device_blocks(pkg, oracle, stream_kind):
    """two sessions of the same size with DIFFERENT inputs, back to back: the second gets the first one's device block back from the pool with the
    first one's records still in it, so any kernel that ran ahead of its upload would produce the first session's values.  Run with the context on
    its own stream and on torch's default stream -- the legacy NULL stream (hipStream_t 0), where "no stream" and "the stream" look alike."""
    fid, n = 0, 1 << 18
    e = pkg.Engine(fid, device=0) if stream_kind == "own" else pkg.Engine(fid, device=0, stream=torch.cuda.current_stream().cuda_stream)
    for seed in (11, 22, 33):
        _, keys, sh = _inputs(fid, n, seed=seed, tile_from=3000 + seed)
        de, out = _run_two_party(e, n, keys, sh)
        ode, want = _oracle_two_party(oracle, fid, n, keys, sh)
        for party in (0, 1):
            assert np.array_equal(de[party], ode[party]), "seed %d: d||e of party %d" % (seed, party)
            assert np.array_equal(out[party], want[party]), "seed %d: result of party %d" % (seed, party)
    e.close()


def test_hostmul_empty_batch(engs):
    e = engs[0]
    z = np.zeros(0, dtype=np.uint64)
    s = e.hostmul_begin(0, z, z, z, z, z, z)
    assert e.hostmul_poll_de(s) == 0
    e.hostmul_wait_de(s)
    e.hostmul_finish(s, 0, np.array([1, 0, 0, 0], dtype=np.uint64), z, z)


def test_hostmul_unaligned_and_aliased_operands(engs, oracle):
    """a Rust Vec<ScalarShare> is 8-byte aligned, nothing more; batch_mul(&a, &a) passes the same vector twice (circuit_mul_throughput.rs:30)"""
    fid, n = 0, 40000
    e = engs[fid]
    _, keys, sh = _inputs(fid, n, seed=77, tile_from=2000)
    def shift(a):                       # the same words at an address that is 8 mod 16
        buf = np.zeros(a.size + 3, dtype=np.uint64)
        off = 1 if (buf.ctypes.data % 16) == 0 else 2
        v = buf[off:off + a.size]; v[:] = a
        assert v.ctypes.data % 16 == 8
        return v
    for party in (0, 1):
        x = shift(sh["x"][party]); a = shift(sh["a"][party]); b = shift(sh["b"][party]); c = shift(sh["c"][party])
        de = shift(np.zeros(8 * n, dtype=np.uint64)); out = shift(np.zeros(8 * n, dtype=np.uint64))
        peer = shift(oracle.beaver_mask(fid, sh["x"][1 - party], sh["x"][1 - party], sh["a"][1 - party], sh["b"][1 - party]))
        s = e.hostmul_begin(n, x, x, a, b, c, de)                  # x * x
        e.hostmul_finish(s, party, keys[party], peer, out)
        my_de, want = oracle.batch_mul_9pass_local(fid, party, keys[party], sh["x"][party], sh["x"][party], sh["a"][party], sh["b"][party],
                                                   sh["c"][party], np.ascontiguousarray(peer))
        assert np.array_equal(de, my_de) and np.array_equal(out, want)


def test_hostmul_preregistered_and_pinned_buffers(pkg, engs, oracle):
    """buffers the caller pinned itself (arkmpc_host_register) and buffers from arkmpc_host_alloc run the same pipeline with no pinning inside"""
    fid, n = 0, 50000
    e = engs[fid]
    lib = pkg.load_library()
    _, keys, sh = _inputs(fid, n, seed=91, tile_from=2000)
    party = 1
    names = "xyabc"
    regs = []
    for k in names:
        arr = sh[k][party]
        assert lib.arkmpc_host_register(ctypes.c_void_p(arr.ctypes.data), ctypes.c_size_t(arr.nbytes)) == 0
        assert lib.arkmpc_host_register(ctypes.c_void_p(arr.ctypes.data), ctypes.c_size_t(arr.nbytes)) == 0      # twice = OK
        regs.append(arr)
    ptrs = []
    def pinned(nwords):
        q = ctypes.c_void_p()
        assert lib.arkmpc_host_alloc(ctypes.c_size_t(8 * nwords), ctypes.byref(q)) == 0
        ptrs.append(q)
        return np.ctypeslib.as_array(ctypes.cast(q, ctypes.POINTER(ctypes.c_uint64)), shape=(nwords,))
    de, out, peer = pinned(8 * n), pinned(8 * n), pinned(8 * n)
    peer[:] = oracle.beaver_mask(fid, sh["x"][0], sh["y"][0], sh["a"][0], sh["b"][0])
    s = e.hostmul_begin(n, *(sh[k][party] for k in names), de)
    e.hostmul_finish(s, party, keys[party], peer, out)
    my_de, want = oracle.batch_mul_9pass_local(fid, party, keys[party], *(sh[k][party] for k in names), np.array(peer))
    assert np.array_equal(de, my_de) and np.array_equal(out, want)
    for arr in regs:
        assert lib.arkmpc_host_unregister(ctypes.c_void_p(arr.ctypes.data)) == 0
    del de, out, peer
    for q in ptrs:
        assert lib.arkmpc_host_free(q) == 0


def test_hostmul_error_paths_end_the_session(pkg, engs):
    fid, n = 0, 1000
    e = engs[fid]
    lib = pkg.load_library()
    _, keys, sh = _inputs(fid, n, seed=5)
    de = np.zeros(8 * n, dtype=np.uint64); out = np.zeros(8 * n, dtype=np.uint64)
    s = ctypes.c_void_p()
    # a null operand is a status, not a crash, and no session is created
    rc = lib.arkmpc_hostmul_begin(e.h, ctypes.c_size_t(n), None, ctypes.c_void_p(sh["y"][0].ctypes.data), ctypes.c_void_p(sh["a"][0].ctypes.data),
                                  ctypes.c_void_p(sh["b"][0].ctypes.data), ctypes.c_void_p(sh["c"][0].ctypes.data), ctypes.c_void_p(de.ctypes.data), ctypes.byref(s))
    assert rc == -1 and not s.value
    # a bad party id in phase 2 returns BAD_ARG and still ends the session (nothing leaks, the context stays usable)
    s = e.hostmul_begin(n, sh["x"][0], sh["y"][0], sh["a"][0], sh["b"][0], sh["c"][0], de)
    with pytest.raises(pkg.ArkMpcError):
        e.hostmul_finish(s, 2, keys[0], de, out)
    # abort after phase 1
    s = e.hostmul_begin(n, sh["x"][0], sh["y"][0], sh["a"][0], sh["b"][0], sh["c"][0], de)
    e.hostmul_abort(s)
    s = e.hostmul_begin(n, sh["x"][0], sh["y"][0], sh["a"][0], sh["b"][0], sh["c"][0], de)
    e.hostmul_finish(s, 0, keys[0], de, out)


def test_hostmul_two_contexts_two_threads(pkg, oracle):
    """the two parties as two host threads with a context each (the shape of execute_mock_mpc, test_helpers.rs): each waits for its own
    payload, hands it to the peer over a host 'link' (a barrier), finishes with the peer's"""
    fid, n = 0, 1 << 18
    _, keys, sh = _inputs(fid, n, seed=123, tile_from=4096)
    es = [pkg.Engine(fid, device=0) for _ in (0, 1)]
    de = [np.zeros(8 * n, dtype=np.uint64) for _ in (0, 1)]
    out = [np.zeros(8 * n, dtype=np.uint64) for _ in (0, 1)]
    bar = threading.Barrier(2)
    errs = []

    def party(p):
        try:
            torch.cuda.set_device(0)
            s = es[p].hostmul_begin(n, sh["x"][p], sh["y"][p], sh["a"][p], sh["b"][p], sh["c"][p], de[p])
            es[p].hostmul_wait_de(s)
            bar.wait(timeout=60)
            es[p].hostmul_finish(s, p, keys[p], de[1 - p], out[p])
        except Exception as ex:      # noqa: BLE001
            errs.append(ex)
            try:
                bar.abort()
            except Exception:        # noqa: BLE001
                pass

    th = [threading.Thread(target=party, args=(p,)) for p in (0, 1)]
    for t in th: t.start()
    for t in th: t.join()
    assert not errs, errs
    ode, want = _oracle_two_party(oracle, fid, n, keys, sh)
    for p in (0, 1):
        assert np.array_equal(de[p], ode[p]) and np.array_equal(out[p], want[p])
    for e in es:
        e.close()


def test_hostmul_config2_all_2p20_gates_bitexact(pkg, oracle):
    """BASELINE config 2 from host memory: 2^20 Beaver muls over BN254 Fr, seeded uniform data generated with the engine, brought to the
    host as a Rust caller would hold it, run through the streaming sessions -- every gate == the oracle's literal 9-pass batch_mul."""
    import test_gpu_fullsize as fs
    fid, n = 0, 1 << 20
    e = fs._eng(pkg, fid)
    vals, shd, key, keys = fs._setup(e, n, 0xA11CE002)
    torch.cuda.synchronize()
    sh = {k: (fs._host(shd[k][0]), fs._host(shd[k][1])) for k in "xyabc"}
    del shd, vals
    torch.cuda.empty_cache()
    de, out = _run_two_party(e, n, keys, sh)
    ode, want = _oracle_two_party(oracle, fid, n, keys, sh)
    for p in (0, 1):
        bad_de = np.nonzero((de[p].reshape(2 * n, 4) != ode[p].reshape(2 * n, 4)).any(axis=1))[0]
        assert bad_de.size == 0, "party %d d||e: %d of %d scalars differ, first at %d, last at %d" % (p, bad_de.size, 2 * n, bad_de[0], bad_de[-1])
        bad = np.nonzero((out[p].reshape(n, 8) != want[p].reshape(n, 8)).any(axis=1))[0]
        assert bad.size == 0, "party %d: %d of %d gates differ, first at %d" % (p, bad.size, n, bad[0])
    e.close()


# ---- host-buffer contexts (arkmpc_ctx_set_host_buffers): large elementwise calls on buffers the caller has pinned run the streamed staging
# ---- of csrc/arkmpc_internal.hpp (last input + kernels + downloads pipelined chunk by chunk); pageable or small ones the whole-batch staging
def _tile(arr, words, n):
    m = arr.size // words
    return np.ascontiguousarray(np.tile(arr.reshape(m, words), (-(-n // m), 1))[:n].reshape(-1))


class _PinnedArena:
    """numpy arrays in memory pinned through the C ABI (arkmpc_host_alloc), so that host-buffer calls take the streamed path"""

    def __init__(self, pkg):
        self.lib = pkg.load_library()
        self.ptrs = []

    def zeros(self, nwords, dtype=np.uint64):
        q = ctypes.c_void_p()
        nbytes = max(16, nwords * np.dtype(dtype).itemsize)
        assert self.lib.arkmpc_host_alloc(ctypes.c_size_t(nbytes), ctypes.byref(q)) == 0
        self.ptrs.append(q)
        ct = ctypes.c_uint64 if dtype == np.uint64 else ctypes.c_uint8
        a = np.ctypeslib.as_array(ctypes.cast(q, ctypes.POINTER(ct)), shape=(nwords,))
        a[:] = 0
        return a

    def copy(self, arr):
        a = self.zeros(arr.size, arr.dtype)
        a[:] = arr
        return a

    def free(self):
        """Deliberately NOT returning the memory: numpy views of it may still be referenced (pytest keeps locals of a frame around for its
        reports), and touching a view of freed pinned memory is a segfault in the test process, not a test failure.  The process exit frees it."""
        self.ptrs = []


@pytest.mark.parametrize("fid", [0, 1])
@pytest.mark.parametrize("pinned", [True, False])
@pytest.mark.parametrize("n", [150001])
def test_host_buffer_mode_large_ops_vs_oracle(pkg, oracle, fid, n, pinned):
    """every elementwise entry point at a size above the streaming threshold (ragged last chunk), host buffers, == oracle: pinned buffers take
    the three-stream pipeline, pageable numpy buffers the whole-batch staging"""
    e = pkg.Engine(fid, device=0, host_buffers=True)
    arena = _PinnedArena(pkg)
    z = (lambda cnt, w: arena.zeros(cnt * w)) if pinned else (lambda cnt, w: np.zeros(cnt * w, dtype=np.uint64))
    H = arena.copy if pinned else (lambda arr: arr)
    base = 2048
    a = H(_tile(mont_array(fid, mixed_values(fid, base, 1)), 4, n))
    b = H(_tile(mont_array(fid, rand_values(fid, base - 1, 2)), 4, n))         # a different period: pairs differ along the batch
    for name, ora in (("scalar_add", oracle.scalar_add), ("scalar_sub", oracle.scalar_sub), ("scalar_mul", oracle.scalar_mul), ("open_combine", oracle.open_combine)):
        out = z(n, 4); getattr(e, name)(n, a, b, out)
        assert np.array_equal(out, ora(fid, a, b)), name
    out = z(n, 4); e.scalar_neg(n, a, out); assert np.array_equal(out, oracle.scalar_neg(fid, a))
    out = z(n, 4); e.scalar_to_canonical(n, a, out); assert np.array_equal(out, oracle.to_canonical(fid, a))
    outb = arena.zeros(32 * n, np.uint8) if pinned else np.zeros(32 * n, dtype=np.uint8)
    e.scalar_to_bytes_be(n, a, outb); assert np.array_equal(outb, oracle.to_bytes_be(fid, a))
    _, keys, sh = _inputs(fid, n, seed=300 + fid, tile_from=base)
    sh = {k: (H(v[0]), H(v[1])) for k, v in sh.items()}
    sa, sb = sh["x"][0], sh["y"][1]
    out = z(n, 8); e.share_add(n, sa, sb, out); assert np.array_equal(out, oracle.share_add(fid, sa, sb))
    out = z(n, 8); e.share_sub(n, sa, sb, out); assert np.array_equal(out, oracle.share_sub(fid, sa, sb))
    out = z(n, 8); e.share_neg(n, sa, out); assert np.array_equal(out, oracle.share_neg(fid, sa))
    out = z(n, 8); e.share_mul_public(n, sa, b, out); assert np.array_equal(out, oracle.share_mul_public(fid, sa, b))
    for party in (0, 1):
        out = z(n, 8); e.share_add_public(n, party, keys[party], sa, b, out)
        assert np.array_equal(out, oracle.share_add_public(fid, party, keys[party], sa, b))
        out = z(n, 8); e.share_sub_public(n, party, keys[party], sa, b, out)
        assert np.array_equal(out, oracle.share_add_public(fid, party, keys[party], sa, b, sub=True))
    out = z(n, 4); e.share_extract(n, sa, out); assert np.array_equal(out.reshape(-1, 4), sa.reshape(-1, 8)[:, :4])
    s_col, m_col = z(n, 4), z(n, 4); e.share_split(n, sa, s_col, m_col)
    assert np.array_equal(s_col.reshape(-1, 4), sa.reshape(-1, 8)[:, :4]) and np.array_equal(m_col.reshape(-1, 4), sa.reshape(-1, 8)[:, 4:])
    out = z(n, 8); e.share_join(n, s_col, m_col, out); assert np.array_equal(out, sa)
    # Beaver multiplication through the two staged calls (a, b and d||e go up again for K2+K3: the sessions above avoid that)
    ode, want = _oracle_two_party(oracle, fid, n, keys, sh)
    opened = oracle.open_combine(fid, ode[0], ode[1])
    ode_h = [H(ode[0]), H(ode[1])]
    for party in (0, 1):
        de = z(2 * n, 4); e.beaver_mask(n, sh["x"][party], sh["y"][party], sh["a"][party], sh["b"][party], de)
        assert np.array_equal(de, ode[party])
        out = z(n, 8); e.beaver_finish_fused(n, party, keys[party], ode_h[party], ode_h[1 - party], sh["a"][party], sh["b"][party], sh["c"][party], out)
        assert np.array_equal(out, want[party])
        out = z(n, 8); e.beaver_finish(n, party, keys[party], H(opened[:4 * n].copy()), H(opened[4 * n:].copy()), sh["a"][party], sh["b"][party], sh["c"][party], out)
        assert np.array_equal(out, want[party])
    # batch open + MAC-check shares
    mine = [H(np.ascontiguousarray(sh["x"][p].reshape(-1, 8)[:, :4].reshape(-1))) for p in (0, 1)]
    for party in (0, 1):
        o, c = z(n, 4), z(n, 4)
        e.open_and_mac_check(n, keys[party], sh["x"][party], mine[1 - party], o, c)
        o_ref = oracle.open_combine(fid, np.array(mine[party]), np.array(mine[1 - party]))
        c_ref = oracle.mac_check_shares(fid, keys[party], o_ref, np.array(sh["x"][party]))
        assert np.array_equal(o, o_ref) and np.array_equal(c, c_ref)
        c2 = z(n, 4); e.mac_check_shares(n, keys[party], H(o_ref), sh["x"][party], c2)
        assert np.array_equal(c2, c_ref)
    e.close()
    del a, b, sh, sa, sb, mine, ode_h, de, out, o, c, c2, s_col, m_col, outb
    arena.free()


@pytest.mark.parametrize("pinned", [True, False])
@pytest.mark.parametrize("stream_kind", ["own", "torch_default"])
def test_host_buffer_mode_in_place_and_repeated_operands(pkg, oracle, pinned, stream_kind):
    """out aliases an input (ScalarResult ops are often written in place by callers), and one vector passed twice; two rounds with different
    values so that the second round's kernels would see the first round's data in the arena if they ran ahead of their uploads; with the context on
    its own stream and on torch's default (NULL) stream"""
    fid, n = 0, 300000
    e = pkg.Engine(fid, device=0, host_buffers=True, stream=None if stream_kind == "own" else torch.cuda.current_stream().cuda_stream)
    arena = _PinnedArena(pkg)
    for seed in (9, 10):
        a0 = _tile(mont_array(fid, mixed_values(fid, 1000, seed)[::-1 if seed & 1 else 1]), 4, n)
        want = oracle.scalar_mul(fid, a0, a0)
        a = arena.copy(a0) if pinned else a0
        out = arena.zeros(4 * n) if pinned else np.zeros(4 * n, dtype=np.uint64)
        e.scalar_mul(n, a, a, out)
        assert np.array_equal(out, want)
        buf = arena.copy(a0) if pinned else a0.copy()
        e.scalar_mul(n, buf, buf, buf)
        assert np.array_equal(buf, want)
    e.close()
    arena.free()


@pytest.mark.parametrize("pinned", [False, True])
def test_hostmul_soak_random_sizes_two_threads(pkg, oracle, pinned):
    """(pinned: the same with every vector in memory the caller pinned -- sessions of 4096 gates and more then run as zero-copy kernels, the
    smaller ones through the copy pipeline, in one interleaved stream of sessions on the same contexts.)
    300 sessions of random sizes (1 ... 40000 gates, so both the single-chunk and the chunked schedules, pinned and unpinned buffer sizes) from
    two host threads with a context each, the parties' payloads crossing between the threads; every result against the oracle.  Leaks of
    events, pins or device blocks would show as errors or as a growing pool; ordering bugs as wrong words."""
    import random
    fid = 0
    rng = random.Random(4242)
    base_n = 40000
    _, keys, sh = _inputs(fid, base_n, seed=777, tile_from=2500)
    arena = _PinnedArena(pkg)
    if pinned:
        sh = {k: (arena.copy(v[0]), arena.copy(v[1])) for k, v in sh.items()}
    es = [pkg.Engine(fid, device=0) for _ in (0, 1)]
    zc_before = _zc_count(pkg, *es)
    sizes = [rng.choice([1, 2, 63, 255, 256, 257, 1000, 4095, 4096, 4097, 16384, 16385, 33000, base_n]) for _ in range(150)]
    offs = [rng.randrange(0, base_n - n + 1) for n in sizes]
    bar = threading.Barrier(2)
    errs = []
    de = [[None] * len(sizes), [None] * len(sizes)]
    out = [[None] * len(sizes), [None] * len(sizes)]

    def party(p):
        try:
            torch.cuda.set_device(0)
            for k, (n, o) in enumerate(zip(sizes, offs)):
                sl = lambda a: a[8 * o: 8 * (o + n)]
                de[p][k] = np.zeros(8 * n, dtype=np.uint64); out[p][k] = np.zeros(8 * n, dtype=np.uint64)
                s = es[p].hostmul_begin(n, sl(sh["x"][p]), sl(sh["y"][p]), sl(sh["a"][p]), sl(sh["b"][p]), sl(sh["c"][p]), de[p][k])
                es[p].hostmul_wait_de(s)
                bar.wait(timeout=60)
                es[p].hostmul_finish(s, p, keys[p], de[1 - p][k], out[p][k])
                bar.wait(timeout=60)                       # the peer must have finished reading my payload before the buffers go away
        except Exception as ex:      # noqa: BLE001
            errs.append(repr(ex))
            try:
                bar.abort()
            except Exception:        # noqa: BLE001
                pass

    # pinned: the payload / result vectors are slices of two pinned pools per party (checked session by session below, so each session's
    # slice is copied out right after the run -- later sessions overwrite the pool)
    pool = [[arena.zeros(8 * base_n), arena.zeros(8 * base_n)] for _ in (0, 1)] if pinned else None
    if pinned:                                       # keep every session's words: run the sessions one by one per index, copying the slices out
        keep_de = [[None] * len(sizes), [None] * len(sizes)]
        keep_out = [[None] * len(sizes), [None] * len(sizes)]

        def party(p):                                # noqa: F811
            try:
                torch.cuda.set_device(0)
                for k, (n, o) in enumerate(zip(sizes, offs)):
                    sl = lambda a: a[8 * o: 8 * (o + n)]
                    d_, o_ = pool[p][0][8 * o: 8 * (o + n)], pool[p][1][8 * o: 8 * (o + n)]
                    d_.fill(0); o_.fill(0)
                    de[p][k] = d_
                    s = es[p].hostmul_begin(n, sl(sh["x"][p]), sl(sh["y"][p]), sl(sh["a"][p]), sl(sh["b"][p]), sl(sh["c"][p]), d_)
                    es[p].hostmul_wait_de(s)
                    bar.wait(timeout=60)
                    es[p].hostmul_finish(s, p, keys[p], de[1 - p][k], o_)
                    keep_de[p][k] = d_.copy(); keep_out[p][k] = o_.copy()
                    bar.wait(timeout=60)
            except Exception as ex:      # noqa: BLE001
                errs.append(repr(ex))
                try:
                    bar.abort()
                except Exception:        # noqa: BLE001
                    pass
    th = [threading.Thread(target=party, args=(p,)) for p in (0, 1)]
    for t in th: t.start()
    for t in th: t.join()
    assert not errs, errs[:2]
    if pinned:
        de, out = keep_de, keep_out
        zc_after = _zc_count(pkg, *es)
        big = sum(1 for n in sizes if n >= 4096)
        assert (zc_after[0] - zc_before[0], zc_after[1] - zc_before[1]) == (2 * big, 2 * big)
    for k, (n, o) in enumerate(zip(sizes, offs)):
        sub = {nm: (np.ascontiguousarray(sh[nm][0][8 * o: 8 * (o + n)]), np.ascontiguousarray(sh[nm][1][8 * o: 8 * (o + n)])) for nm in "xyabc"}
        ode, want = _oracle_two_party(oracle, fid, n, keys, sub)
        for p in (0, 1):
            assert np.array_equal(de[p][k], ode[p]) and np.array_equal(out[p][k], want[p]), (k, n, o, p)
    for e in es:
        e.close()
    arena.free()


# ---- zero-copy phases: vectors the caller pinned are read and written IN PLACE by k_hostmul_mask / k_hostmul_finish (no copy commands);
# ---- each phase falls back to the DMA pipeline on its own.  The test hook counts the phases that ran zero-copy, so these tests also
# ---- pin WHICH path produced the (identical) words.
_TRACKED = []           # the module's shared engines (fixture `engs`): their per-context counters are what the path assertions read


def _zc_count(pkg, *extra):
    """phases that ran as zero-copy kernels so far (arkmpc_ctx_get_stats), summed over the module's shared engines and `extra`"""
    tot = [0, 0]
    for e in list(_TRACKED) + list(extra):
        z = e.stats()["hostmul_zero_copy_phases"]
        tot[0] += z[0]; tot[1] += z[1]
    return tot[0], tot[1]


def _two_party_on(eng, n, keys, sh, place):
    """_run_two_party with every vector put where `place(name, array)` says (pinned arena, pageable copy, shifted ...)"""
    H = [{k: place(k, sh[k][p]) for k in "xyabc"} for p in (0, 1)]
    de = [place("de", np.zeros(8 * n, dtype=np.uint64)) for _ in (0, 1)]
    out = [place("out", np.zeros(8 * n, dtype=np.uint64)) for _ in (0, 1)]
    ses = [eng.hostmul_begin(n, H[p]["x"], H[p]["y"], H[p]["a"], H[p]["b"], H[p]["c"], de[p]) for p in (0, 1)]
    for p in (0, 1):
        eng.hostmul_wait_de(ses[p])
        assert eng.hostmul_poll_de(ses[p]) == n
    for p in (0, 1):
        eng.hostmul_finish(ses[p], p, keys[p], de[1 - p], out[p])
    return de, out


@pytest.mark.parametrize("fid", [0, 1, 2])
@pytest.mark.parametrize("n", [4096, 4097, 70001])
def test_hostmul_zero_copy_bitexact_vs_oracle(pkg, engs, oracle, fid, n):
    """all eight vectors of both parties pinned: both phases run as kernels on the caller's memory; ragged last tile (n % 256 != 0)"""
    arena = _PinnedArena(pkg)
    _, keys, sh = _inputs(fid, n, seed=8100 + n, tile_from=3000)
    before = _zc_count(pkg)
    de, out = _two_party_on(engs[fid], n, keys, sh, lambda k, a: arena.copy(a))
    after = _zc_count(pkg)
    assert (after[0] - before[0], after[1] - before[1]) == (2, 2), "both phases of both parties must have run zero-copy"
    ode, want = _oracle_two_party(oracle, fid, n, keys, sh)
    for party in (0, 1):
        assert np.array_equal(de[party], ode[party]), "d||e of party %d" % party
        assert np.array_equal(out[party], want[party]), "result of party %d" % party
    arena.free()


def test_hostmul_zero_copy_two_launches_per_phase(pkg, oracle):
    """2^20 + 4173 gates: two launches per phase (one per 2^20 gates), the second one short and ragged; torch's default stream"""
    fid, n = 0, (1 << 20) + 4173
    arena = _PinnedArena(pkg)
    e = pkg.Engine(fid, device=0, stream=torch.cuda.current_stream().cuda_stream)
    _, keys, sh = _inputs(fid, n, seed=8200, tile_from=4001)
    before = _zc_count(pkg, e)
    de, out = _two_party_on(e, n, keys, sh, lambda k, a: arena.copy(a))
    after = _zc_count(pkg, e)
    assert (after[0] - before[0], after[1] - before[1]) == (2, 2)
    ode, want = _oracle_two_party(oracle, fid, n, keys, sh)
    for party in (0, 1):
        assert np.array_equal(de[party], ode[party]) and np.array_equal(out[party], want[party]), "party %d" % party
    e.close()
    arena.free()


@pytest.mark.parametrize("pageable", ["phase2", "phase1", "c_only", "x_only"])
def test_hostmul_mixed_zero_copy_and_dma_phases(pkg, engs, oracle, pageable):
    """one phase's vectors pageable, the other's pinned: the pinned phase runs zero-copy, the other through the copy pipeline, and they hand
    a, b and d||e to each other in HBM.  c_only: c pageable -> it is uploaded, phase 2 still runs zero-copy on the peer's payload and the result."""
    fid, n = 0, 70001
    arena = _PinnedArena(pkg)
    _, keys, sh = _inputs(fid, n, seed=8300, tile_from=3000)
    loose = {"phase2": ("c", "out"), "phase1": ("x", "y", "a", "b", "de"), "c_only": ("c",), "x_only": ("x",)}[pageable]
    # (de is both party p's phase-1 output and party 1-p's phase-2 input, so "phase1" makes the peer payloads pageable too: phase 2 then is DMA)
    place = lambda k, a: (np.array(a) if k in loose else arena.copy(a))
    before = _zc_count(pkg)
    de, out = _two_party_on(engs[fid], n, keys, sh, place)
    after = _zc_count(pkg)
    got = (after[0] - before[0], after[1] - before[1])
    # "phase1": the payload vectors are pageable; each party's pin on its own d||e ends at _wait_de (the vector is the caller's again), so both
    # parties' phase 2 find a pageable peer payload and go through the copy pipeline
    assert got == {"phase2": (2, 0), "phase1": (0, 0), "c_only": (2, 2), "x_only": (0, 2)}[pageable], got
    ode, want = _oracle_two_party(oracle, fid, n, keys, sh)
    for party in (0, 1):
        assert np.array_equal(de[party], ode[party]) and np.array_equal(out[party], want[party]), "party %d" % party
    arena.free()


def test_hostmul_zero_copy_needs_16_byte_alignment_and_result_may_reuse_an_input(pkg, engs, oracle):
    """pinned vectors at 8 mod 16 (a sub-slice of a pinned Vec) take the copy pipeline -- the kernels move 16-byte quarters --; and a result
    vector that IS the x vector (dead after phase 1) or the c vector (same index, read before written within a tile) is fine in both modes"""
    fid, n = 0, 20000
    e = engs[fid]
    arena = _PinnedArena(pkg)
    _, keys, sh = _inputs(fid, n, seed=8400, tile_from=2000)
    ode, want = _oracle_two_party(oracle, fid, n, keys, sh)

    def shifted(k, a):
        buf = arena.zeros(a.size + 2)
        assert buf.ctypes.data % 16 == 0
        v = buf[1:1 + a.size]; v[:] = a
        return v
    before = _zc_count(pkg)
    de, out = _two_party_on(e, n, keys, sh, shifted)
    assert _zc_count(pkg) == before
    for party in (0, 1):
        assert np.array_equal(de[party], ode[party]) and np.array_equal(out[party], want[party])
    for reuse, pin in (("x", True), ("c", True), ("c", False)):
        for party in (0, 1):
            put = (lambda a: arena.copy(a)) if pin else (lambda a: np.array(a))
            H = {k: put(sh[k][party]) for k in "xyabc"}
            my_de = put(np.zeros(8 * n, dtype=np.uint64)); peer = put(ode[1 - party])
            s = e.hostmul_begin(n, H["x"], H["y"], H["a"], H["b"], H["c"], my_de)
            e.hostmul_finish(s, party, keys[party], peer, H[reuse])
            assert np.array_equal(my_de, ode[party]) and np.array_equal(H[reuse], want[party]), (reuse, pin, party)
    arena.free()


def test_exception_inside_an_entry_point_aborts_instead_of_unwinding(pkg):
    """The boundary's callers are Rust closures: nothing may unwind across the C ABI.  A C++ exception inside an entry point's body (std::bad_alloc
    is the realistic one) must end the process at the guard every entry point holds, with the library's message -- run in a child process."""
    import subprocess, sys, os, signal
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import importlib, sys; sys.path.insert(0, %r); pkg = importlib.import_module('ark-mpc_amd'); e = pkg.Engine(0, device=0); "
            "import ctypes, os; hooks = ctypes.CDLL(os.path.join(os.path.dirname(pkg.lib_path()), 'libarkmpc_testhooks.so')); "
            "print('before', flush=True); rc = hooks.arkmpc_test_throw_inside(e.h); print('after', rc, flush=True)" % root)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert "before" in r.stdout and "after" not in r.stdout
    assert r.returncode == -signal.SIGABRT, r.returncode
    assert "aborting instead of unwinding into the caller" in r.stderr


@pytest.mark.parametrize("how", ["host_alloc", "host_register"])
def test_hostmul_zero_copy_sees_what_the_cpu_wrote_between_sessions(pkg, engs, oracle, how):
    """a caller that keeps its pinned vectors across gates REWRITES them between sessions: the kernels read host memory directly, so any
    line of a previous session still cached on the GPU side would come back stale.  Three sessions on the same eight pinned vectors per party,
    new contents each time, results read by the CPU in between."""
    fid, n = 0, 30000
    e = engs[fid]
    lib = pkg.load_library()
    arena = _PinnedArena(pkg)
    regs = []

    def pinned(nwords):
        if how == "host_alloc":
            return arena.zeros(nwords)
        a = FreshVA.zeros(nwords)             # (addresses in their first registered life: a recycled numpy address would travel by DMA, tests/test_gpu_pinning.py)
        assert lib.arkmpc_host_register(ctypes.c_void_p(a.ctypes.data), ctypes.c_size_t(a.nbytes)) == 0
        regs.append(a)
        return a
    H = [{k: pinned(8 * n) for k in "xyabc"} for _ in (0, 1)]
    de = [pinned(8 * n) for _ in (0, 1)]
    out = [pinned(8 * n) for _ in (0, 1)]
    before = _zc_count(pkg)
    for seed in (1, 2, 3):
        _, keys, sh = _inputs(fid, n, seed=8500 + seed, tile_from=1500 + seed)
        for p in (0, 1):
            for k in "xyabc":
                H[p][k][:] = sh[k][p]
            de[p].fill(seed); out[p].fill(seed)
        ses = [e.hostmul_begin(n, H[p]["x"], H[p]["y"], H[p]["a"], H[p]["b"], H[p]["c"], de[p]) for p in (0, 1)]
        for p in (0, 1):
            e.hostmul_wait_de(ses[p])
        for p in (0, 1):
            e.hostmul_finish(ses[p], p, keys[p], de[1 - p], out[p])
        ode, want = _oracle_two_party(oracle, fid, n, keys, sh)
        for p in (0, 1):
            assert np.array_equal(de[p], ode[p]) and np.array_equal(out[p], want[p]), "seed %d party %d" % (seed, p)
    after = _zc_count(pkg)
    assert (after[0] - before[0], after[1] - before[1]) == (6, 6)
    for a in regs:
        assert lib.arkmpc_host_unregister(ctypes.c_void_p(a.ctypes.data)) == 0
    arena.free()


def test_hostmul_zero_copy_on_slices_of_one_registered_region(pkg, engs, oracle):
    """all eight vectors are sub-ranges of ONE region the caller registered (arkmpc_host_register on a slab it carves its Vecs from): the device-side
    address of a vector is then the region's mapping plus an offset, not a mapping of its own"""
    fid, n = 0, 9000
    e = engs[fid]
    lib = pkg.load_library()
    _, keys, sh = _inputs(fid, n, seed=8600, tile_from=1500)
    slab = FreshVA.zeros(2 * 8 * (8 * n + 8) + 64)               # (an address in its first registered life)
    off0 = (-(slab.ctypes.data // 8)) % 2                        # first word on a 16-byte boundary
    assert lib.arkmpc_host_register(ctypes.c_void_p(slab.ctypes.data), ctypes.c_size_t(slab.nbytes)) == 0
    cur = [off0 + 2]                                             # (not the region's first byte)

    def carve(a):
        v = slab[cur[0]: cur[0] + a.size]; cur[0] += a.size + 8  # 64-byte gaps between the vectors
        v[:] = a
        assert v.ctypes.data % 16 == 0
        return v
    before = _zc_count(pkg)
    de, out = _two_party_on(e, n, keys, sh, lambda k, a: carve(a))
    after = _zc_count(pkg)
    assert (after[0] - before[0], after[1] - before[1]) == (2, 2)
    ode, want = _oracle_two_party(oracle, fid, n, keys, sh)
    for party in (0, 1):
        assert np.array_equal(de[party], ode[party]) and np.array_equal(out[party], want[party]), "party %d" % party
    assert lib.arkmpc_host_unregister(ctypes.c_void_p(slab.ctypes.data)) == 0


def test_hostmul_zero_copy_on_torch_pinned_tensors(pkg, engs, oracle):
    """pinned memory that somebody else allocated (torch's caching host allocator): the library pins nothing, maps nothing of its own, and still
    runs both phases in place"""
    fid, n = 0, 12345
    _, keys, sh = _inputs(fid, n, seed=8700, tile_from=1500)
    keepalive = []

    def place(k, a):
        t = torch.empty(a.size, dtype=torch.int64).pin_memory()
        keepalive.append(t)
        v = t.numpy().view(np.uint64)
        v[:] = a
        return v
    before = _zc_count(pkg)
    de, out = _two_party_on(engs[fid], n, keys, sh, place)
    after = _zc_count(pkg)
    assert (after[0] - before[0], after[1] - before[1]) == (2, 2)
    ode, want = _oracle_two_party(oracle, fid, n, keys, sh)
    for party in (0, 1):
        assert np.array_equal(de[party], ode[party]) and np.array_equal(out[party], want[party]), "party %d" % party


# ---- sessions with the payloads in their wire form (arkmpc_hostmul_begin_wire / _finish_wire): the frame each party would put on a QUIC stream
# ---- (network/quic.rs:303-306) checked byte for byte against the serde_json model (pyref.wire_frame), the results against the oracle
def _wire_two_party(eng, fid, n, keys, sh, place, rids=(100, 2 ** 63 + 5)):
    H = [{k: place(sh[k][p]) for k in "xyabc"} for p in (0, 1)]
    cap = eng.wire_frame_bound(2 * n)
    frames = [place(np.zeros(cap, dtype=np.uint8)) for _ in (0, 1)]
    out = [place(np.zeros(8 * n, dtype=np.uint64)) for _ in (0, 1)]
    ses, lens = [], []
    for p in (0, 1):
        s, ln = eng.hostmul_begin_wire(n, H[p]["x"], H[p]["y"], H[p]["a"], H[p]["b"], H[p]["c"], rids[p], frames[p])
        ses.append(s); lens.append(ln)
    got_rids = [eng.hostmul_finish_wire(ses[p], p, keys[p], frames[1 - p], lens[1 - p], out[p]) for p in (0, 1)]
    return [frames[p][:lens[p]].tobytes() for p in (0, 1)], out, got_rids


@pytest.mark.parametrize("pinned", [False, True])
@pytest.mark.parametrize("fid,n", [(0, 1), (0, 777), (0, 16389), (1, 5000), (2, 4097)])
def test_hostmul_wire_sessions_vs_model_and_oracle(pkg, engs, oracle, fid, n, pinned):
    e = engs[fid]
    arena = _PinnedArena(pkg)
    _, keys, sh = _inputs(fid, n, seed=9000 + n, tile_from=2000)
    rids = (100, 2 ** 63 + 5)
    before = _zc_count(pkg)
    frames, out, got_rids = _wire_two_party(e, fid, n, keys, sh, (lambda a: arena.copy(a)) if pinned else (lambda a: np.array(a)), rids)
    after = _zc_count(pkg)
    ode, want = _oracle_two_party(oracle, fid, n, keys, sh)
    for p in (0, 1):
        model = pyref.wire_frame("ScalarBatch", rids[p], pyref.wire_scalar_records(fid, from_mont_array(fid, ode[p])))
        assert frames[p] == model, "frame of party %d" % p
        assert np.array_equal(out[p], want[p]), "result of party %d" % p
        assert got_rids[p] == rids[1 - p]
    if pinned and n >= 4096:
        assert (after[0] - before[0], after[1] - before[1]) == (2, 2)           # the phases themselves still run in place on the pinned vectors
    arena.free()


def test_hostmul_wire_session_rejects_bad_peer_frames(pkg, engs, oracle):
    """a peer frame with one scalar too few, with a scalar >= the modulus, with a syntax error, of the wrong variant: ARKMPC_ERR_BAD_ARG, the session
    ends, and the context runs the next session normally"""
    fid, n = 0, 300
    e = engs[fid]
    p_mod = pyref.P[fid]
    _, keys, sh = _inputs(fid, n, seed=9100)
    ode, want = _oracle_two_party(oracle, fid, n, keys, sh)
    good_vals = from_mont_array(fid, ode[1])
    recs = pyref.wire_scalar_records(fid, good_vals)
    bad = {
        "one scalar too few": pyref.wire_frame("ScalarBatch", 1, recs[:-1]),
        "one scalar too many": pyref.wire_frame("ScalarBatch", 1, recs + recs[:1]),
        "scalar >= modulus": pyref.wire_frame("ScalarBatch", 1, recs[:5] + [int(p_mod).to_bytes(32, "little")] + recs[6:]),
        "syntax": pyref.wire_frame("ScalarBatch", 1, recs).replace(b"],[", b"],,[", 1),
        "variant": pyref.wire_frame("PointBatch", 1, recs),
    }
    bad["syntax"] = bad["syntax"][8:]
    import struct
    bad["syntax"] = struct.pack("<Q", len(bad["syntax"])) + bad["syntax"]
    cap = e.wire_frame_bound(2 * n)
    for why, fr in bad.items():
        frame = np.zeros(cap, dtype=np.uint8); out = np.zeros(8 * n, dtype=np.uint64)
        s, ln = e.hostmul_begin_wire(n, sh["x"][0], sh["y"][0], sh["a"][0], sh["b"][0], sh["c"][0], 7, frame)
        peer = np.frombuffer(fr, dtype=np.uint8).copy()
        with pytest.raises(pkg.ArkMpcError):
            e.hostmul_finish_wire(s, 0, keys[0], peer, len(peer), out)
        assert not out.any(), why
    # too small a frame buffer is refused before anything is enqueued
    with pytest.raises(pkg.ArkMpcError):
        e.hostmul_begin_wire(n, sh["x"][0], sh["y"][0], sh["a"][0], sh["b"][0], sh["c"][0], 7, np.zeros(cap - 1, dtype=np.uint8))
    frame = np.zeros(cap, dtype=np.uint8); out = np.zeros(8 * n, dtype=np.uint64)
    s, ln = e.hostmul_begin_wire(n, sh["x"][0], sh["y"][0], sh["a"][0], sh["b"][0], sh["c"][0], 7, frame)
    peer = np.frombuffer(pyref.wire_frame("ScalarBatch", 9, recs), dtype=np.uint8).copy()
    assert e.hostmul_finish_wire(s, 0, keys[0], peer, len(peer), out) == 9
    assert np.array_equal(out, want[0])


def test_hostmul_wire_session_reads_what_serde_would(pkg, engs, oracle):
    """the peer's frame need not be in serde_json::to_vec's compact form: anything serde_json::from_slice reads as the same NetworkOutbound -- whitespace
    between tokens, the two fields in the other order, an unknown field -- finishes the session with the same words"""
    import json, struct
    fid, n = 0, 200
    e = engs[fid]
    _, keys, sh = _inputs(fid, n, seed=9300)
    ode, want = _oracle_two_party(oracle, fid, n, keys, sh)
    recs = [list(r) for r in pyref.wire_scalar_records(fid, from_mont_array(fid, ode[1]))]
    shapes = {
        "whitespace": json.dumps({"result_id": 3, "payload": {"ScalarBatch": recs}}, indent=1),
        "fields swapped": json.dumps({"payload": {"ScalarBatch": recs}, "result_id": 3}, separators=(",", ":")),
        "unknown field": json.dumps({"result_id": 3, "note": {"a": [1, 2, {"b": None}]}, "payload": {"ScalarBatch": recs}}, separators=(",", ":")),
    }
    cap = e.wire_frame_bound(2 * n)
    for why, body in shapes.items():
        frame = np.zeros(cap, dtype=np.uint8); out = np.zeros(8 * n, dtype=np.uint64)
        s, ln = e.hostmul_begin_wire(n, sh["x"][0], sh["y"][0], sh["a"][0], sh["b"][0], sh["c"][0], 7, frame)
        b = body.encode()
        peer = np.frombuffer(struct.pack("<Q", len(b)) + b, dtype=np.uint8).copy()
        assert e.hostmul_finish_wire(s, 0, keys[0], peer, len(peer), out) == 3, why
        assert np.array_equal(out, want[0]), why


def test_host_register_of_a_vector_a_session_has_pinned(pkg, engs, oracle):
    """the caller registers a vector WHILE a session of the library has it pinned in place: the registration becomes a reference on that pin, survives
    the session's end, and arkmpc_host_unregister gives it back (ROCm itself would take the second registration and drop both at the first unregister)"""
    fid, n = 0, 40000
    e = engs[fid]
    lib = pkg.load_library()
    _, keys, sh = _inputs(fid, n, seed=9400, tile_from=2000)
    ode, want = _oracle_two_party(oracle, fid, n, keys, sh)
    x = np.array(sh["x"][0])
    reg = lambda a: lib.arkmpc_host_register(ctypes.c_void_p(a.ctypes.data), ctypes.c_size_t(a.nbytes))
    unreg = lambda a: lib.arkmpc_host_unregister(ctypes.c_void_p(a.ctypes.data))
    for rnd in range(2):
        de = np.zeros(8 * n, dtype=np.uint64); out = np.zeros(8 * n, dtype=np.uint64)
        s = e.hostmul_begin(n, x, sh["y"][0], sh["a"][0], sh["b"][0], sh["c"][0], de)
        if rnd == 0:
            assert reg(x) == 0 and reg(x) == 0                     # while the session holds the pin; twice = still one reference
        e.hostmul_finish(s, 0, keys[0], ode[1], out)
        assert np.array_equal(de, ode[0]) and np.array_equal(out, want[0])
    assert unreg(x) == 0
    assert unreg(x) != 0                                           # no longer registered: a status, not a crash
    de = np.zeros(8 * n, dtype=np.uint64); out = np.zeros(8 * n, dtype=np.uint64)
    s = e.hostmul_begin(n, x, sh["y"][0], sh["a"][0], sh["b"][0], sh["c"][0], de)
    e.hostmul_finish(s, 0, keys[0], ode[1], out)
    assert np.array_equal(out, want[0])


def test_host_alloc_recycles_blocks(pkg):
    """arkmpc_host_alloc / _free keep freed pinned blocks on a free list by size class (the runtime's own alloc + free of 64 MiB cost 16 ms): the same
    block comes back for the same class, another class gets another block, double frees and foreign pointers are a status, trim empties the list"""
    lib = pkg.load_library()
    assert lib.arkmpc_host_trim() == 0
    def alloc(nbytes):
        q = ctypes.c_void_p()
        assert lib.arkmpc_host_alloc(ctypes.c_size_t(nbytes), ctypes.byref(q)) == 0 and q.value
        return q
    a = alloc(3 << 20)
    ctypes.memset(a, 0x5A, 3 << 20)
    assert lib.arkmpc_host_free(a) == 0
    b = alloc((3 << 20) - 4097)                                   # same 1 MiB class
    assert b.value == a.value
    c = alloc(3 << 20)                                            # the list is empty again: a new block
    assert c.value != b.value
    d = alloc(5 << 20)
    assert d.value not in (b.value, c.value)
    assert lib.arkmpc_host_free(b) == 0 and lib.arkmpc_host_free(b) != 0          # double free
    assert lib.arkmpc_host_free(ctypes.c_void_p(b.value + 64)) != 0              # not a block
    buf = np.zeros(16, dtype=np.uint64)
    assert lib.arkmpc_host_free(ctypes.c_void_p(buf.ctypes.data)) != 0
    assert lib.arkmpc_host_free(c) == 0 and lib.arkmpc_host_free(d) == 0
    assert lib.arkmpc_host_trim() == 0
    assert lib.arkmpc_host_free(c) != 0                                           # gone with the trim
    small = alloc(1000)
    assert lib.arkmpc_host_free(small) == 0 and alloc(65536).value == small.value     # 64 KiB classes below 1 MiB
