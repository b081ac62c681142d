"""GPU parity of the round-5 host-link work, every word against the CPU oracle (authenticated_scalar.rs:848-879):
  * range / placement forms of the streaming session (arkmpc_hostmul_begin_range, _finish_async, _end): each vector pageable, pinned or
    RESIDENT in HBM on its own -- the shape of a circuit gate (x, y and the result resident, triples in host memory, fabric.rs:894-915);
  * device memory a session holds, per path (arkmpc_ctx_get_stats);
  * streaming sessions over the multi-device group (arkmpc_group_hostmul_*): members sharing device 0 on the one-GPU box, ragged n, 2^20,
    == a single-context session == the oracle; the distinct-device variant skips loudly when the box has one GPU;
  * group transfers from pinned host memory (in-place import kernel for split columns);
  * asynchronous batch imports (arkmpc_batch_from_host_async / _acquire / _host_release)."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

from test_gpu_stream import _inputs, _oracle_two_party, _PinnedArena, _run_two_party

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _dev(arr):
    return torch.from_numpy(arr.view(np.int64)).cuda()


def _host(t):
    return t.cpu().numpy().view(np.uint64)


@pytest.fixture(scope="module")
def eng0(pkg):
    return pkg.Engine(0, device=0)


# ---- placements -------------------------------------------------------------------------------------------------------------------------
PLACEMENTS = {
    # name: where (x, y | a, b, c | out_de, peer_de | out) live
    "circuit_gate_pinned_triples": ("dev", "pin", "pin", "dev"),
    "circuit_gate_pageable_triples": ("dev", "page", "page", "dev"),
    "all_resident": ("dev", "dev", "dev", "dev"),
    "resident_operands_host_result": ("dev", "pin", "pin", "pin"),
    "host_operands_resident_triples": ("pin", "dev", "pin", "pin"),
    "device_link_payload": ("pin", "pin", "dev", "pin"),
    "pageable_operands_resident_result": ("page", "page", "page", "dev"),
}


@pytest.mark.parametrize("n", [777, 4096, 70001])
@pytest.mark.parametrize("name", sorted(PLACEMENTS))
def test_hostmul_every_vector_placed_on_its_own(pkg, eng0, oracle, name, n):
    fid = 0
    _, keys, sh = _inputs(fid, n, seed=9100 + n, tile_from=3000)
    ode, want = _oracle_two_party(oracle, fid, n, keys, sh)
    arena = _PinnedArena(pkg)
    keep = []

    def put(where, arr):
        if where == "dev":
            t = _dev(arr); keep.append(t); return t
        if where == "pin":
            return arena.copy(arr)
        return arr.copy()

    w_xy, w_tri, w_pay, w_out = PLACEMENTS[name]
    H = [{"x": put(w_xy, sh["x"][p]), "y": put(w_xy, sh["y"][p]), "a": put(w_tri, sh["a"][p]), "b": put(w_tri, sh["b"][p]), "c": put(w_tri, sh["c"][p])} for p in (0, 1)]
    de = [put(w_pay, np.zeros(8 * n, dtype=np.uint64)) for _ in (0, 1)]
    out = [put(w_out, np.zeros(8 * n, dtype=np.uint64)) for _ in (0, 1)]

    def off(buf, words):                       # element offset into a numpy array or a tensor
        return buf[words:]

    ses = [eng0.hostmul_begin_range(n, H[p]["x"], H[p]["y"], H[p]["a"], H[p]["b"], H[p]["c"], de[p], off(de[p], 4 * n)) for p in (0, 1)]
    for p in (0, 1):
        eng0.hostmul_wait_de(ses[p])
    for p in (0, 1):
        eng0.hostmul_finish_async(ses[p], p, keys[p], de[1 - p], off(de[1 - p], 4 * n), out[p])
    for p in (0, 1):
        eng0.hostmul_end(ses[p])
    got = lambda b: _host(b) if hasattr(b, "data_ptr") else b
    for p in (0, 1):
        assert np.array_equal(got(de[p]), ode[p]), (name, "d||e", p)
        assert np.array_equal(got(out[p]), want[p]), (name, "result", p)
    arena.free()


def test_hostmul_rejects_misaligned_device_vectors_and_double_finish(pkg, eng0):
    n = 5000
    _, keys, sh = _inputs(0, n, seed=9200, tile_from=1000)
    x = _dev(np.concatenate([sh["x"][0], np.zeros(8, dtype=np.uint64)]))
    de, out = np.zeros(8 * n, dtype=np.uint64), np.zeros(8 * n, dtype=np.uint64)
    with pytest.raises(pkg.ArkMpcError, match="16-byte"):
        eng0.hostmul_begin_range(n, x[1:], sh["y"][0], sh["a"][0], sh["b"][0], sh["c"][0], de, de[4 * n:])
    with pytest.raises(pkg.ArkMpcError, match="unknown session flag"):
        eng0.hostmul_begin_range(n, sh["x"][0], sh["y"][0], sh["a"][0], sh["b"][0], sh["c"][0], de, de[4 * n:], flags=6)
    s = eng0.hostmul_begin_range(n, sh["x"][0], sh["y"][0], sh["a"][0], sh["b"][0], sh["c"][0], de, de[4 * n:])
    peer = np.zeros(8 * n, dtype=np.uint64)
    eng0.hostmul_finish_async(s, 0, keys[0], peer, peer[4 * n:], out)
    with pytest.raises(pkg.ArkMpcError, match="already"):
        eng0.hostmul_finish_async(s, 0, keys[0], peer, peer[4 * n:], out)
    eng0.hostmul_end(s)                         # the failed call left the session open: it still ends cleanly


def test_hostmul_session_holds_only_what_its_path_touches(pkg, oracle):
    """zero-copy phases stash a, b and d||e: 192 B per gate (it was 512 whatever the path); the copy pipeline stages everything: 512"""
    fid, n = 0, 1 << 16
    e = pkg.Engine(fid, device=0)
    _, keys, sh = _inputs(fid, n, seed=9300, tile_from=2000)
    arena = _PinnedArena(pkg)
    P = {k: (arena.copy(v[0]), arena.copy(v[1])) for k, v in sh.items()}
    de = [arena.zeros(8 * n) for _ in (0, 1)]
    out = [arena.zeros(8 * n) for _ in (0, 1)]
    s = e.hostmul_begin(n, P["x"][0], P["y"][0], P["a"][0], P["b"][0], P["c"][0], de[0])
    ode, want = _oracle_two_party(oracle, fid, n, keys, sh)
    e.hostmul_finish(s, 0, keys[0], ode[1], out[0])             # (the peer's payload from the oracle: pageable -> phase 2 stages the peer, c and the result)
    st = e.stats()
    assert st["hostmul_zero_copy_phases"] == (1, 0) and st["hostmul_copy_phases"] == (0, 1)
    assert st["hostmul_device_bytes_last"] == n * (192 + 192)   # stash + c, peer, result staging
    peer = arena.copy(ode[1])
    s = e.hostmul_begin(n, P["x"][0], P["y"][0], P["a"][0], P["b"][0], P["c"][0], de[0])
    e.hostmul_finish(s, 0, keys[0], peer, out[0])
    st = e.stats()
    assert st["hostmul_zero_copy_phases"] == (2, 1)
    assert st["hostmul_device_bytes_last"] == n * 192, st
    assert np.array_equal(out[0], want[0]) and np.array_equal(de[0], ode[0])
    # resident triples: nothing to stash but d||e
    ta, tb, tc = (_dev(sh[k][0]) for k in "abc")
    s = e.hostmul_begin(n, P["x"][0], P["y"][0], ta, tb, tc, de[0])
    e.hostmul_finish(s, 0, keys[0], peer, out[0])
    assert e.stats()["hostmul_device_bytes_last"] == n * 64
    assert np.array_equal(out[0], want[0])
    # pageable everything: the copy pipeline stages all eight vectors
    s = e.hostmul_begin(n, sh["x"][0], sh["y"][0], sh["a"][0], sh["b"][0], sh["c"][0], np.zeros(8 * n, dtype=np.uint64))
    o2 = np.zeros(8 * n, dtype=np.uint64)
    e.hostmul_finish(s, 0, keys[0], ode[1], o2)
    assert e.stats()["hostmul_device_bytes_last"] == n * 512 and e.stats()["hostmul_device_bytes_peak"] == n * 512
    assert np.array_equal(o2, want[0])
    e.close()
    arena.free()


def test_hostmul_payload_vector_is_the_callers_again_after_wait_de(pkg, eng0, oracle):
    """round-4 advisor finding: the copy path kept out_de registered until _finish.  After _wait_de the vector may be dropped; a NEW array that
    lands on the same address must be pinned afresh by the next session, not mistaken for an old registration."""
    fid, n = 0, 40000
    _, keys, sh = _inputs(fid, n, seed=9400, tile_from=2500)
    ode, want = _oracle_two_party(oracle, fid, n, keys, sh)
    for _ in range(6):
        de = np.zeros(8 * n, dtype=np.uint64)
        s = eng0.hostmul_begin(n, sh["x"][0], sh["y"][0], sh["a"][0], sh["b"][0], sh["c"][0], de)
        eng0.hostmul_wait_de(s)
        assert np.array_equal(de, ode[0])
        attr_free = not _is_registered(pkg, de)
        del de                                                   # freed before phase 2
        out = np.zeros(8 * n, dtype=np.uint64)
        eng0.hostmul_finish(s, 0, keys[0], ode[1], out)
        assert np.array_equal(out, want[0])
        assert attr_free, "out_de was still registered after _wait_de"


def _is_registered(pkg, arr):
    """True if the HIP runtime tracks the array's memory (pinned / registered)"""
    hip = ctypes.CDLL("libamdhip64.so")
    buf = (ctypes.c_uint8 * 256)()
    rc = hip.hipPointerGetAttributes(buf, ctypes.c_void_p(arr.ctypes.data))
    if rc != 0:
        hip.hipGetLastError()
        return False
    return ctypes.cast(buf, ctypes.POINTER(ctypes.c_int))[0] != 0          # hipMemoryTypeUnregistered = 0


# ---- group sessions ----------------------------------------------------------------------------------------------------------------------
def _group_two_party(pkg, devs, n, keys, place, sh, fid=0, poll=False):
    grp = [pkg.Group(fid, devs) for _ in (0, 1)]
    H = [{k: place(sh[k][p]) for k in "xyabc"} for p in (0, 1)]
    de = [place(np.zeros(8 * n, dtype=np.uint64)) for _ in (0, 1)]
    out = [place(np.zeros(8 * n, dtype=np.uint64)) for _ in (0, 1)]
    ses = [grp[p].hostmul_begin(n, H[p]["x"], H[p]["y"], H[p]["a"], H[p]["b"], H[p]["c"], de[p]) for p in (0, 1)]
    for p in (0, 1):
        if poll:
            assert 0 <= grp[p].hostmul_poll_de(ses[p]) <= n
        grp[p].hostmul_wait_de(ses[p])
        assert grp[p].hostmul_poll_de(ses[p]) == n
    for p in (0, 1):
        grp[p].hostmul_finish(ses[p], p, keys[p], de[1 - p], out[p])
    stats = [[grp[p].member_stats(m) for m in range(len(devs))] for p in (0, 1)]
    for g in grp:
        g.close()
    return de, out, stats


@pytest.mark.parametrize("pinned", [True, False])
@pytest.mark.parametrize("n,G", [(100003, 4), (70001, 3), (16384, 8), (5, 8), (1, 2), (0, 3), (4097, 1)])
def test_group_sessions_oversubscribed_equal_single_context_and_oracle(pkg, eng0, oracle, n, G, pinned):
    fid = 0
    _, keys, sh = _inputs(fid, max(n, 1), seed=9500 + n, tile_from=3000)
    if n == 0:
        sh = {k: (v[0][:0], v[1][:0]) for k, v in sh.items()}
    arena = _PinnedArena(pkg)
    place = (lambda a: arena.copy(a)) if pinned else (lambda a: a.copy())
    de, out, stats = _group_two_party(pkg, [0] * G, n, keys, place, sh, poll=True)
    if n:
        ode, want = _oracle_two_party(oracle, fid, n, keys, sh)
        one_de, one_out = _run_two_party(eng0, n, keys, sh)
        for p in (0, 1):
            assert np.array_equal(de[p], ode[p]) and np.array_equal(out[p], want[p]), ("oracle", p)
            assert np.array_equal(de[p], one_de[p]) and np.array_equal(out[p], one_out[p]), ("single context", p)
    if pinned and n:
        # members whose range reaches the zero-copy threshold ran both phases in place
        for p in (0, 1):
            for m in range(G):
                cnt = (n * (m + 1)) // G - (n * m) // G
                z = stats[p][m]["hostmul_zero_copy_phases"]
                assert z == ((1, 1) if cnt >= 4096 else (0, 0)), (p, m, cnt, z)
    arena.free()


def test_group_sessions_config2_size_all_gates(pkg, oracle):
    """2^20 + 333 gates over 4 members sharing device 0, caller-pinned vectors: every word of d||e and of the result of both parties"""
    fid, n, G = 0, (1 << 20) + 333, 4
    _, keys, sh = _inputs(fid, n, seed=9600, tile_from=4001)
    arena = _PinnedArena(pkg)
    de, out, stats = _group_two_party(pkg, [0] * G, n, keys, lambda a: arena.copy(a), sh)
    ode, want = _oracle_two_party(oracle, fid, n, keys, sh)
    for p in (0, 1):
        assert np.array_equal(de[p], ode[p]) and np.array_equal(out[p], want[p]), p
        for m in range(G):
            assert stats[p][m]["hostmul_zero_copy_phases"] == (1, 1)
            assert stats[p][m]["hostmul_device_bytes_peak"] <= 192 * (n // G + 1)
    arena.free()


def test_group_sessions_on_distinct_devices(pkg, oracle):
    """the same on two PHYSICAL GPUs: each member's kernels address the shared pinned vectors over its own PCIe link"""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs: this box has %d -- the distinct-device group session has NOT run here" % torch.cuda.device_count())
    fid, n = 0, 200003
    _, keys, sh = _inputs(fid, n, seed=9700, tile_from=3000)
    arena = _PinnedArena(pkg)
    for place in ((lambda a: arena.copy(a)), (lambda a: a.copy())):
        de, out, _ = _group_two_party(pkg, [0, 1], n, keys, place, sh)
        ode, want = _oracle_two_party(oracle, fid, n, keys, sh)
        for p in (0, 1):
            assert np.array_equal(de[p], ode[p]) and np.array_equal(out[p], want[p]), p
    arena.free()


def test_group_session_errors_end_the_session(pkg):
    n, G = 9000, 3
    _, keys, sh = _inputs(0, n, seed=9800, tile_from=1000)
    g = pkg.Group(0, [0] * G)
    de, out = np.zeros(8 * n, dtype=np.uint64), np.zeros(8 * n, dtype=np.uint64)
    with pytest.raises(pkg.ArkMpcError):
        g.hostmul_begin(n, sh["x"][0], None, sh["a"][0], sh["b"][0], sh["c"][0], de)
    s = g.hostmul_begin(n, sh["x"][0], sh["y"][0], sh["a"][0], sh["b"][0], sh["c"][0], de)
    with pytest.raises(pkg.ArkMpcError, match="party_id"):
        g.hostmul_finish(s, 2, keys[0], de, out)             # ends the session
    s = g.hostmul_begin(n, sh["x"][0], sh["y"][0], sh["a"][0], sh["b"][0], sh["c"][0], de)
    g.hostmul_abort(s)
    s = g.hostmul_begin(n, sh["x"][0], sh["y"][0], sh["a"][0], sh["b"][0], sh["c"][0], de)    # and the group is still usable
    g.hostmul_wait_de(s)
    g.hostmul_finish(s, 0, keys[0], np.zeros(8 * n, dtype=np.uint64), out)
    g.close()


@pytest.mark.parametrize("pinned", [True, False])
@pytest.mark.parametrize("layout", ["aos", "split"])
def test_group_host_transfers_from_pinned_and_pageable_vectors(pkg, layout, pinned):
    """arkmpc_group_shares_from_host / _to_host / scatter_h2d / gather_d2h: pinned vectors go as concurrent true DMAs (split: the in-place import
    kernel), pageable ones are pinned for the call; the words are the same"""
    n, G = 150001, 4
    rng = np.random.default_rng(5)
    rec = rng.integers(0, 1 << 63, size=8 * n, dtype=np.uint64)
    arena = _PinnedArena(pkg)
    src = arena.copy(rec) if pinned else rec.copy()
    g = pkg.Group(0, [0] * G)
    L = pkg.Group.SPLIT if layout == "split" else pkg.Group.AOS
    sh = g.malloc(n, 2 if layout == "split" else 1, 4 if layout == "split" else 8)
    g.shares_from_host(L, n, src, sh)
    for m in range(G):                                       # each shard against the records of its range
        lo, cnt = g.shard_range(n, m)
        t = torch.empty(8 * cnt, dtype=torch.int64, device="cuda")
        pkg.load_library().arkmpc_memcpy_d2d(g.member_ctx(m), ctypes.c_void_p(t.data_ptr()), ctypes.c_void_p(sh[m]), ctypes.c_size_t(64 * cnt))
        g.sync()
        got = _host(t)
        want = rec[8 * lo: 8 * (lo + cnt)].reshape(cnt, 8)
        if layout == "split":
            assert np.array_equal(got[:4 * cnt].reshape(cnt, 4), want[:, :4]) and np.array_equal(got[4 * cnt:].reshape(cnt, 4), want[:, 4:]), m
        else:
            assert np.array_equal(got.reshape(cnt, 8), want), m
    back = arena.zeros(8 * n) if pinned else np.zeros(8 * n, dtype=np.uint64)
    g.shares_to_host(L, n, sh, back)
    assert np.array_equal(back, rec)
    g.free(sh)
    g.close()
    arena.free()


# ---- asynchronous batch imports ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("how", ["pinned", "pageable", "pinned_unaligned", "small", "device"])
@pytest.mark.parametrize("layout", ["split", "aos"])
def test_batch_from_host_async_equals_the_blocking_import(pkg, layout, how):
    n = 300 if how == "small" else 200003
    e = pkg.Engine(0, device=0)
    rng = np.random.default_rng(11)
    rec = rng.integers(0, 1 << 63, size=8 * n + 8, dtype=np.uint64)
    arena = _PinnedArena(pkg)
    if how == "pinned":
        src = arena.copy(rec)[:8 * n]
    elif how == "pinned_unaligned":
        src = arena.copy(rec)[1:8 * n + 1]                        # 8-byte aligned only, like a Rust Vec may be
    elif how == "device":
        keep = _dev(rec); src = keep[:8 * n]
    else:
        src = rec[:8 * n].copy()
    want_rec = _host(src) if how == "device" else np.array(src, copy=True)
    L = e.SPLIT if layout == "split" else e.AOS
    before = e.stats()
    b = e.batch_from_host(e.SCALAR_SHARE, L, n, src, asynchronous=True)
    e.batch_acquire(b)
    got = np.zeros(8 * n, dtype=np.uint64)
    e.batch_to_host(b, got)
    e.batch_host_release(b)
    e.batch_host_release(b)                                       # idempotent
    after = e.stats()
    assert np.array_equal(got, want_rec)
    went_async = after["batch_async_imports"] - before["batch_async_imports"]
    # a pageable vector is registered in place for the import (then DMA + split from a staging block: no kernel addresses a vector the library
    # registered itself) unless ARKMPC_PIN_IN_PLACE=0 / ARKMPC_NO_PIN=1: then it takes the blocking import
    in_place = os.environ.get("ARKMPC_PIN_IN_PLACE") != "0" and os.environ.get("ARKMPC_NO_PIN") != "1"
    assert went_async == (0 if how == "small" or (how == "pageable" and not in_place) else 1)
    assert after["batch_blocking_imports"] - before["batch_blocking_imports"] == 1 - went_async
    # the columns themselves (split): share column then MAC column
    if layout == "split":
        sp, mp, stride = e.batch_ptrs(b)
        assert stride == 4 and mp == sp + 32 * n
        t = torch.empty(8 * n, dtype=torch.int64, device="cuda")
        e.call("memcpy_d2d", t, sp, ("size", 64 * n)); e.sync()
        cols = _host(t)
        assert np.array_equal(cols[:4 * n].reshape(n, 4), want_rec.reshape(n, 8)[:, :4]) and np.array_equal(cols[4 * n:].reshape(n, 4), want_rec.reshape(n, 8)[:, 4:])
    e.batch_destroy(b)
    if how in ("pageable",):
        assert not _is_registered(pkg, src), "the import's pin must end with _host_release"
    # destroy without release: the pin ends there
    if how == "pageable":
        b2 = e.batch_from_host(e.SCALAR_SHARE, L, n, src, asynchronous=True)
        e.batch_destroy(b2)
        assert not _is_registered(pkg, src)
    e.close()
    arena.free()


def test_prefetched_triples_feed_a_chain_of_gates(pkg, oracle):
    """a depth-4 chain z <- z * y with the operands resident in split columns and every gate's triples imported asynchronously from host memory
    one gate AHEAD of their use (what MpcFabric::next_triple_batch does): each gate's d||e and result == the oracle on the same records"""
    fid, n, depth = 0, 50001, 4
    e = pkg.Engine(fid, device=0)
    arena = _PinnedArena(pkg)
    _, keys, sh = _inputs(fid, n, seed=9900, tile_from=2000)
    trip = []
    for k in range(depth):
        _, _, t = _inputs(fid, n, seed=9900 + 31 * (k + 1), tile_from=2000)
        trip.append({nm: (arena.copy(t[nm][0]) if k % 2 else t[nm][0].copy(), arena.copy(t[nm][1]) if k % 2 else t[nm][1].copy()) for nm in "abc"})
    S = e.SCALAR_SHARE

    def imp(k, p):
        return [e.batch_from_host(S, e.SPLIT, n, trip[k][nm][p], asynchronous=True) for nm in "abc"]

    cur = [[e.batch_from_host(S, e.SPLIT, n, sh[nm][p]) for nm in "xy"] for p in (0, 1)]        # z, y per party (resident)
    host_z = [sh["x"][0].copy(), sh["x"][1].copy()]
    nxt = [imp(0, p) for p in (0, 1)]
    for k in range(depth):
        tri = nxt
        if k + 1 < depth:
            nxt = [imp(k + 1, p) for p in (0, 1)]                                               # gate k+1's triples go up behind gate k
        de = [torch.empty(8 * n, dtype=torch.int64, device="cuda") for _ in (0, 1)]
        outb = [e.batch_from_host(S, e.SPLIT, n, np.zeros(8 * n, dtype=np.uint64)) for _ in (0, 1)]
        for p in (0, 1):
            for b in tri[p]:
                e.batch_acquire(b)
            (zs, zm, st), (ys, ym, _), (as_, am, _), (bs, bm, _) = (e.batch_ptrs(q) for q in (cur[p][0], cur[p][1], tri[p][0], tri[p][1]))
            e.beaver_mask_v(n, zs, st, ys, st, as_, st, bs, st, de[p])
        for p in (0, 1):
            (as_, am, st), (bs, bm, _), (cs, cm, _), (os_, om, _) = (e.batch_ptrs(q) for q in (tri[p][0], tri[p][1], tri[p][2], outb[p]))
            e.beaver_finish_fused_v(n, p, keys[p], de[p], de[1 - p], as_, am, st, bs, bm, st, cs, cm, st, os_, om, st)
        e.sync()
        t_h = {nm: (trip[k][nm][0], trip[k][nm][1]) for nm in "abc"}
        t_h["x"] = (host_z[0], host_z[1]); t_h["y"] = sh["y"]
        ode, want = _oracle_two_party(oracle, fid, n, keys, t_h)
        for p in (0, 1):
            got = np.zeros(8 * n, dtype=np.uint64)
            e.batch_to_host(outb[p], got)
            assert np.array_equal(_host(de[p]), ode[p]) and np.array_equal(got, want[p]), (k, p)
            host_z[p] = got
            for b in tri[p]:
                e.batch_host_release(b); e.batch_destroy(b)
            e.batch_destroy(cur[p][0])
            cur[p][0] = outb[p]
    e.close()
    arena.free()


def _where_differs(name, got, ref, orc):
    """which words of a group session's vector differ from the single-context session's, where they lie relative to 4 KiB pages, and which of
    the two the oracle sides with"""
    idx = np.nonzero(got != ref)[0]
    if not idx.size:
        return "%s equal" % name
    runs = np.split(idx, np.nonzero(np.diff(idx) > 1)[0] + 1)
    base = got.ctypes.data
    return "%s: %d of %d words differ in %d run(s) %s (byte address mod 4096 of the runs' ends: %s); group == oracle: %s, one context == oracle: %s, group words there all zero: %s" % (
        name, idx.size, got.size, len(runs), [(int(r[0]), int(r[-1])) for r in runs[:6]],
        [((base + 8 * int(r[0])) % 4096, (base + 8 * int(r[-1]) + 8) % 4096) for r in runs[:6]],
        bool(np.array_equal(got, orc)), bool(np.array_equal(ref, orc)), bool(np.all(got[idx] == 0)))


def test_group_sessions_soak_random_sizes_members_and_placements(pkg, oracle):
    """24 group sessions of random size (1 ... 70 000 gates), member count (1 ... 8, all on device 0) and vector placement (pinned pools /
    pageable / freshly allocated per session), two parties interleaved, each against a single-context session on the same records (itself
    checked against the oracle elsewhere).  Leaks of events, pins or device blocks show as errors or a growing pool; ordering bugs as wrong words."""
    import random
    fid = 0
    rng = random.Random(515)
    base = 70000
    eng0 = pkg.Engine(fid, device=0)             # its own context: hostmul_device_bytes_peak is per context, and the module's has run larger sessions
    _, keys, sh = _inputs(fid, base, seed=9950, tile_from=2500)
    arena = _PinnedArena(pkg)
    pinned = {k: (arena.copy(v[0]), arena.copy(v[1])) for k, v in sh.items()}
    pool_de = [arena.zeros(8 * base), arena.zeros(8 * base)]
    pool_out = [arena.zeros(8 * base), arena.zeros(8 * base)]
    groups = {}
    for it in range(24):
        n = rng.choice([1, 2, 255, 257, 4095, 4096, 4097, 9000, 16385, 33000, 65536, base])
        G = rng.choice([1, 2, 3, 5, 8])
        how = rng.choice(["pinned", "pageable", "mixed"])
        o = rng.randrange(0, base - n + 1)
        sl = lambda a: a[8 * o: 8 * (o + n)]
        if G not in groups:
            groups[G] = [pkg.Group(fid, [0] * G) for _ in (0, 1)]
        grp = groups[G]
        src = pinned if how != "pageable" else sh
        H = [{k: (sl(src[k][p]) if how != "mixed" or k in "xa" else sl(sh[k][p]).copy()) for k in "xyabc"} for p in (0, 1)]
        if how == "pageable":
            de = [np.zeros(8 * n, dtype=np.uint64) for _ in (0, 1)]; out = [np.zeros(8 * n, dtype=np.uint64) for _ in (0, 1)]
        else:
            de = [pool_de[p][:8 * n] for p in (0, 1)]; out = [pool_out[p][:8 * n] for p in (0, 1)]
            for a_ in de + out:
                a_.fill(0)
        ses = [grp[p].hostmul_begin(n, H[p]["x"], H[p]["y"], H[p]["a"], H[p]["b"], H[p]["c"], de[p]) for p in (0, 1)]
        for p in (0, 1):
            grp[p].hostmul_wait_de(ses[p])
        for p in (0, 1):
            grp[p].hostmul_finish(ses[p], p, keys[p], de[1 - p], out[p])
        sub = {nm: (np.ascontiguousarray(sl(sh[nm][0])), np.ascontiguousarray(sl(sh[nm][1]))) for nm in "xyabc"}
        one_de, one_out = _run_two_party(eng0, n, keys, sub)
        for p in (0, 1):
            if not (np.array_equal(de[p], one_de[p]) and np.array_equal(out[p], one_out[p])):
                ode, want = _oracle_two_party(oracle, fid, n, keys, sub)
                pytest.fail("session %r, party %d: %s" % ((it, n, G, how, o), p, "; ".join(
                    _where_differs(nm, got, ref, orc) for nm, got, ref, orc in (("d||e", de[p], one_de[p], ode[p]), ("result", out[p], one_out[p], want[p])))))
    for pair in groups.values():
        for g in pair:
            g.close()
    st = eng0.stats()
    assert st["hostmul_device_bytes_peak"] <= 512 * base
    eng0.close()
    arena.free()


def _rerun_in_child(env_extra, k_expr, files=("test_gpu_stream.py", "test_gpu_group_stream.py")):
    """the named tests again in a child process with the environment switch set (the switches are read once per process).  One attempt, 300 s:
    a child is a handful of tests whose behaviour the switch changes, not a second run of the suite."""
    env = dict(os.environ, **env_extra)
    here = os.path.dirname(os.path.abspath(__file__))
    cmd = [sys.executable, "-m", "pytest", "-q", "-m", "gpu", "-x", "-p", "no:cacheprovider"] + [os.path.join(here, f) for f in files] + ["-k", k_expr]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and " passed" in r.stdout, "%r:\n%s" % (env_extra, (r.stdout[-3000:] + r.stderr[-1500:]).strip())


# one pageable session per field at five chunks, one group session over three members, the asynchronous import of a pageable vector in both layouts,
# a chain of gates fed by read-ahead triples -- the calls whose path the registration switches change.  (The soaks are not re-run: ARKMPC_SOAK=full
# tools/soak_loop.sh does that, with any switch.)
_PAGEABLE_TESTS = ("(test_hostmul_bitexact_vs_oracle and 70001) or (oversubscribed and 70001 and False) or (batch_from_host_async and pageable) "
                   "or prefetched_triples or (mixed_zero_copy and phase2)")


def test_child_never_registering_callers_vectors_produces_the_same_words():
    """ARKMPC_PIN_IN_PLACE=0: the library never registers a caller's vector; pageable vectors travel as the runtime's own pageable copies.  (The
    variable is read once per process, hence the child.)"""
    _rerun_in_child({"ARKMPC_PIN_IN_PLACE": "0"}, _PAGEABLE_TESTS)


def test_group_members_in_threads_for_pageable_vectors():
    """ARKMPC_GROUP_THREADS=1: the member calls that touch a pageable vector run in one host thread per member (what the group does by itself
    when its members sit on distinct devices -- pageable copies block their caller, so this is what keeps N links busy without registering the
    caller's memory).  The variable is read once per process: the pageable group sessions and transfers and the C99 group caller run again in a
    child with it set, members sharing device 0."""
    _rerun_in_child({"ARKMPC_GROUP_THREADS": "1", "ARKMPC_PIN_IN_PLACE": "0"},      # (unregistered vectors of every size: registered ones only enqueue and need no threads)
                    "(oversubscribed and False) or (host_transfers and False) or (group_oversubscribed_equals_single_context and (100003 or 5-8))",
                    files=("test_gpu_group_stream.py", "test_group.py"))
