"""Run as a script in a FRESH process by tests/test_gpu_pinning.py (the pin registry is process-wide: which path a session takes depends on
what was registered earlier in the process).  Every scenario drives streaming sessions of the C ABI (arkmpc_hostmul_*) on vectors placed at
never-used addresses (helpers.FreshVA), compares every word of d||e and of the result records with the oracle, and reports which path ran
(arkmpc_ctx_get_stats).  Prints ONE JSON object: {scenario: {...}}.

The rule under test (csrc/arkmpc_internal.hpp PinRegistry / Place::zc): a kernel addresses a host vector in place only if
the CALLER holds it in pinned memory and its addresses are in their FIRST registered life; everything else travels by DMA."""
import ctypes
import importlib
import json
import os
import sys
import threading
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

import oracle_api                                           # noqa: E402
from helpers import FreshVA                                 # noqa: E402
from test_gpu_stream import _inputs, _oracle_two_party      # noqa: E402

FID, N = 0, 1 << 16            # 4 MiB record vectors: above the 1 MiB floor below which the library does not pin a pageable vector


def main():
    pkg = importlib.import_module("ark-mpc_amd")
    lib = pkg.load_library()
    ora = oracle_api.load()
    _, keys, sh = _inputs(FID, N, seed=9900, tile_from=2000)
    ode, want = _oracle_two_party(ora, FID, N, keys, sh)
    reg = lambda a: lib.arkmpc_host_register(ctypes.c_void_p(a.ctypes.data), ctypes.c_size_t(a.nbytes))
    unreg = lambda a: lib.arkmpc_host_unregister(ctypes.c_void_p(a.ctypes.data))
    res = {}

    def vectors():
        """party 0's eight vectors at fresh addresses: x, y, a, b, c, the peer's payload; zeroed d||e and result"""
        v = {k: FreshVA.copy(sh[k][0]) for k in "xyabc"}
        v["peer"] = FreshVA.copy(ode[1])
        v["de"], v["out"] = FreshVA.zeros(8 * N), FreshVA.zeros(8 * N)
        return v

    def fill(v):
        for k in "xyabc":
            v[k][:] = sh[k][0]
        v["peer"][:] = ode[1]
        v["de"].fill(0); v["out"].fill(0)

    def session(eng, v, between=None):
        """one party-0 session; returns (zero-copy phases, copy phases, refused count) it added and whether every word matched the oracle"""
        s0 = eng.stats()
        s = eng.hostmul_begin(N, v["x"], v["y"], v["a"], v["b"], v["c"], v["de"])
        if between:
            between()
        eng.hostmul_finish(s, 0, keys[0], v["peer"], v["out"])
        s1 = eng.stats()
        d = lambda k: [int(b - a) for a, b in zip(s0[k], s1[k])]
        return {"zero_copy": d("hostmul_zero_copy_phases"), "copy": d("hostmul_copy_phases"),
                "refused": s1["zc_refused_reused_address"] - s0["zc_refused_reused_address"],
                "exact": bool(np.array_equal(v["de"], ode[0]) and np.array_equal(v["out"], want[0]))}

    eng = pkg.Engine(FID, device=0)

    # A / B: first registered life -> kernels in place; the SAME vectors registered a second time -> DMA only
    v = vectors()
    assert all(reg(a) == 0 for a in v.values())
    res["first_life"] = session(eng, v)
    assert all(unreg(a) == 0 for a in v.values())
    assert all(reg(a) == 0 for a in v.values())
    fill(v)
    res["second_life_same_pages"] = session(eng, v)
    assert all(unreg(a) == 0 for a in v.values())

    # registering the same pointer again is harmless -- unless the second call asks for more than the first one pinned
    grow = FreshVA.zeros(1 << 18)
    assert reg(grow[: 1 << 17]) == 0 and reg(grow[: 1 << 17]) == 0 and reg(grow[: 1 << 16]) == 0
    assert reg(grow) != 0
    assert unreg(grow[: 1 << 17]) == 0 and reg(grow) == 0 and unreg(grow) == 0

    # C: register -> session -> unregister -> free -> the same addresses handed out again with NEW pages -> register -> session
    v = vectors()
    assert all(reg(a) == 0 for a in v.values())
    first = session(eng, v)
    assert all(unreg(a) == 0 for a in v.values())
    for a in v.values():
        FreshVA.remap(a)
    fill(v)
    assert all(reg(a) == 0 for a in v.values())
    res["second_life_new_pages"] = dict(session(eng, v), first_session=first)
    assert all(unreg(a) == 0 for a in v.values())

    # D: a life as the library's own per-call registration counts too: pageable session first, then the caller registers the vectors
    v = vectors()
    pageable = session(eng, v)
    assert all(reg(a) == 0 for a in v.values())
    fill(v)
    res["library_life_then_caller"] = dict(session(eng, v), pageable_session=pageable)
    assert all(unreg(a) == 0 for a in v.values())

    # E: the caller registers its vectors WHILE a session holds the library's per-call pins on them: the registrations become references on
    # those pins, the ranges are the caller's from then on (still their first registered life), and the NEXT session runs in place
    v = vectors()
    during = session(eng, v, between=lambda: [reg(a) for a in v.values()])
    fill(v)
    res["caller_registers_during_session"] = dict(session(eng, v), session_it_registered_in=during)
    assert all(unreg(a) == 0 for a in v.values())
    assert unreg(v["x"]) != 0                                         # no longer registered: a status, not a crash

    # G: a vector that begins in one registration and ends in another, with unregistered pages between them, is NOT pinned memory: no kernel may
    # reach it through the first range's device alias (that faults: XNACK is off).  The HIP runtime refuses to copy such a range too
    # (hipMemcpyAsync: invalid argument), so the session ends with a STATUS -- and the context keeps working
    region = FreshVA.zeros((12 << 20) // 8)
    lo, hi = region[: (4 << 20) // 8], region[(8 << 20) // 8:]
    assert reg(lo) == 0 and reg(hi) == 0
    v = vectors()
    good_x = v["x"]
    outcomes = {}
    for name, off in (("begins_inside", 3 << 20), ("ends_inside", 6 << 20)):      # [3, 7) MiB starts in `lo`; [6, 10) MiB ends in `hi`
        x = region[off // 8: off // 8 + 8 * N]
        x[:] = sh["x"][0]
        v["x"] = x
        fill(v)
        try:
            outcomes[name] = session(eng, v)
        except pkg.ArkMpcError as ex:
            outcomes[name] = {"status": str(ex)[:160]}
    v["x"] = good_x
    fill(v)
    outcomes["session_after"] = session(eng, v)
    res["vector_straddling_registrations"] = outcomes
    assert unreg(lo) == 0 and unreg(hi) == 0

    # F: pinning is not serialised by the registry: while two threads register 64 MiB vectors (FIRST registrations: ~2 ms each -- the runtime
    # caches a range it has registered before, a second hipHostRegister of the same range returns in 2 us, probes/register_cycle_probe.py),
    # another thread's registry operations on a range that is already held (a reference on an existing entry: no runtime call) stay microseconds
    CYC = 8

    def fresh_big(k):
        out = []
        for _ in range(k):
            b = FreshVA.zeros((64 << 20) // 8)
            b.fill(1)                                                  # touched: the pages exist
            out.append(b)
        return out

    held = FreshVA.zeros((2 << 20) // 8); held.fill(1)
    assert reg(held) == 0
    t_small, stop = [], threading.Event()

    def churn(vs, times):
        for b in vs:
            t0 = time.perf_counter(); assert reg(b) == 0; times.append(time.perf_counter() - t0)
            assert unreg(b) == 0

    def small_ops():
        k = 0
        while not stop.is_set():
            sub = held[512 * (1 + k % 64):]                            # a pointer inside the held entry, not the one it was registered under
            t0 = time.perf_counter()
            lib.arkmpc_host_register(ctypes.c_void_p(sub.ctypes.data), ctypes.c_size_t(4096))
            lib.arkmpc_host_unregister(ctypes.c_void_p(sub.ctypes.data))
            t_small.append(time.perf_counter() - t0)
            k += 1

    alone, together = [], [[], []]
    vs = fresh_big(CYC)
    t0 = time.perf_counter(); churn(vs, alone); one_thread = time.perf_counter() - t0
    for b in vs:
        FreshVA.release(b)
    sets_ = [fresh_big(CYC), fresh_big(CYC)]
    th = threading.Thread(target=small_ops); th.start()
    ths = [threading.Thread(target=churn, args=(sets_[i], together[i])) for i in (0, 1)]
    t0 = time.perf_counter()
    for t in ths: t.start()
    for t in ths: t.join()
    two_threads = time.perf_counter() - t0
    stop.set(); th.join()
    for vs in sets_:
        for b in vs:
            FreshVA.release(b)
    assert unreg(held) == 0
    res["concurrent_pinning"] = {"register_64MiB_ms_median": float(np.median(alone)) * 1e3, "register_64MiB_ms_median_two_threads": float(np.median(together[0] + together[1])) * 1e3,
                                 "one_thread_%d_registrations_ms" % CYC: one_thread * 1e3, "two_threads_%d_registrations_each_ms" % CYC: two_threads * 1e3,
                                 "two_over_one": two_threads / one_thread,
                                 "small_ops": len(t_small), "small_op_us_median": float(np.median(t_small)) * 1e6,
                                 "small_op_us_p90": float(np.percentile(t_small, 90)) * 1e6, "small_op_us_max": float(np.max(t_small)) * 1e6}
    eng.close()
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
