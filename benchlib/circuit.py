"""Circuit leg: a chain of dependent batch_mul gates whose operands are resident in HBM and whose triples come from host memory."""
import time

import numpy as np
import torch

from .common import FID, build_workload, load_oracle, oracle_sample, pcie_calibration, pinned_array


def leg_circuit(pkg, eng, dev, log2n=20, depth=8):
    """A circuit whose operands are RESIDENT and whose triples are not: a chain of `depth` dependent batch_mul gates z <- z * y at 2^log2n gates,
    both parties in-process on this GPU, d||e handed over in HBM (the mock's device link), every gate on FRESH random triples that lie in host
    memory as a PreprocessingPhase hands them over (fabric.rs:894-915 next_triple_batch, offline_prep.rs:65-81) -- not the dummy source's
    one-record shortcut.  192 B of triples per party-gate must cross the host link, which bounds the circuit whatever the kernels do (at 56 GB/s:
    2.9e8 party-gates/s per GPU).  Measured: the product path (arkmpc_batch_from_host_async: in-place import kernel for split columns, gate k+1's
    triples going up behind gate k), the same from pageable vectors (pinned in place per gate, DMA into a staging block, split from HBM), the round-4 path (blocking copy + split pass per
    triple vector, nothing overlapped), and the streaming session with resident operands (x, y and the result in HBM, triples read in place).
    Every gate's d||e and result of both parties is compared with the oracle."""
    lib = pkg.load_library()
    n = 1 << log2n
    cal = pcie_calibration()
    first, truth = build_workload(eng, n, seed=0xA11CE0C0, layout="aos")
    keys = [p.key for p in first]
    S = eng.SCALAR_SHARE
    hold = []                                                   # pinned blocks to give back

    def to_pinned(t):
        arr, q = pinned_array(lib, 8 * n)
        arr[:] = t.cpu().numpy().view(np.uint64)
        hold.append(q)
        return arr

    trip = []                                                   # trip[k][p] = {"a","b","c"} pinned host record vectors
    for k in range(depth):
        ps = first if k == 0 else build_workload(eng, n, seed=0xA11CE0C0 + 101 * k, layout="aos", key_shares=keys)[0]
        trip.append([{nm: to_pinned(getattr(ps[p], nm)) for nm in "abc"} for p in (0, 1)])
    hx = [first[p].x.cpu().numpy().view(np.uint64).copy() for p in (0, 1)]
    hy = [first[p].y.cpu().numpy().view(np.uint64).copy() for p in (0, 1)]
    x_aos = [first[p].x for p in (0, 1)]
    y_aos = [first[p].y for p in (0, 1)]
    del truth
    # resident operands in split columns
    def split_of(t):
        o = torch.empty_like(t)
        eng.share_split(n, t, o[:4 * n], o[4 * n:])
        return o
    x_sp = [split_of(x_aos[p]) for p in (0, 1)]
    y_sp = [split_of(y_aos[p]) for p in (0, 1)]
    z = [[torch.empty(8 * n, dtype=torch.int64, device="cuda") for _ in range(depth)] for _ in (0, 1)]      # every gate's output stays resident for the check
    de = [[torch.empty(8 * n, dtype=torch.int64, device="cuda") for _ in range(depth)] for _ in (0, 1)]
    P = lambda t: t.data_ptr()
    col = 32 * n                                                # byte offset of the MAC column

    def run_batches(source, asynchronous, prefetch, net_ms=0.0):
        """source(k, p) -> {"a","b","c"} host vectors of gate k.  net_ms > 0: every gate's d||e exchange takes that long (the payload is complete,
        the host waits, then K2+K3 is issued) -- the network round a real two-party deployment has between K1 and K2+K3 of every gate"""
        def imp(k, p):
            return [eng.batch_from_host(S, eng.SPLIT, n, source(k, p)[nm], asynchronous=asynchronous) for nm in "abc"]
        t0 = time.perf_counter()
        nxt = [imp(0, p) for p in (0, 1)]
        for k in range(depth):
            tri = nxt
            zin = x_sp if k == 0 else [z[0][k - 1], z[1][k - 1]]
            ptr = [[eng.batch_ptrs(b) for b in tri[p]] for p in (0, 1)]
            for p in (0, 1):
                for b in tri[p]:
                    eng.batch_acquire(b)
                (as_, am, st), (bs, bm, _), _c = ptr[p]
                eng.beaver_mask_v(n, P(zin[p]), 4, P(y_sp[p]), 4, as_, st, bs, st, de[p][k])
            if prefetch and k + 1 < depth:
                nxt = [imp(k + 1, p) for p in (0, 1)]           # gate k+1's triples start on their way under gate k
            if net_ms > 0:
                eng.sync()                                      # the payload is complete ...
                time.sleep(net_ms * 1e-3)                       # ... and crosses the network
            for p in (0, 1):
                (as_, am, st), (bs, bm, _), (cs, cm, _) = ptr[p]
                eng.beaver_finish_fused_v(n, p, keys[p], de[p][k], de[1 - p][k], as_, am, st, bs, bm, st, cs, cm, st, P(z[p][k]), P(z[p][k]) + col, 4)
            for p in (0, 1):
                for b in tri[p]:
                    eng.batch_host_release(b); eng.batch_destroy(b)
            if not prefetch and k + 1 < depth:
                nxt = [imp(k + 1, p) for p in (0, 1)]
        eng.sync()
        return time.perf_counter() - t0

    def run_sessions():
        """the streaming session as a circuit gate: x, y, the payloads and the result resident (AoS records), a, b, c read in place over the link"""
        t0 = time.perf_counter()
        for k in range(depth):
            zin = x_aos if k == 0 else [zs[0][k - 1], zs[1][k - 1]]
            ses = [eng.hostmul_begin_range(n, zin[p], y_aos[p], trip[k][p]["a"], trip[k][p]["b"], trip[k][p]["c"], des[p][k], P(des[p][k]) + 32 * n) for p in (0, 1)]
            for p in (0, 1):
                eng.hostmul_finish_async(ses[p], p, keys[p], des[1 - p][k], P(des[1 - p][k]) + 32 * n, zs[p][k])
            for p in (0, 1):
                eng.hostmul_end(ses[p])
        eng.sync()
        return time.perf_counter() - t0

    pinned_src = lambda k, p: trip[k][p]

    def run_pageable():                                         # NEW pageable vectors for every gate (numpy / Vec memory), made before the clock starts
        fresh = {(k, p): {nm: trip[k][p][nm].copy() for nm in "abc"} for k in range(depth) for p in (0, 1)}
        return run_batches(lambda k, p: fresh[(k, p)], True, True)

    up_bytes = 192 * 2 * n * depth
    floor_ms = up_bytes / cal["h2d_GBps"] / 1e6
    modes = {}
    ok = True

    def record(name, what, fn, reps=3):
        fn()                                                    # warm (pool blocks, events)
        ts = [fn() for _ in range(reps)]
        t = float(np.median(ts))
        modes[name] = {"what": what, "ms": t * 1e3, "ms_per_gate": t * 1e3 / depth, "party_gates_per_s": 2 * n * depth / t,
                       "triple_GBps_over_the_link": up_bytes / t / 1e9, "frac_of_link_floor": floor_ms / (t * 1e3)}

    # the oracle's chain on the same host data (first m gates of every batch; the chain is elementwise)
    ora = load_oracle()
    m = oracle_sample(n)
    sl8 = lambda a_: np.ascontiguousarray(a_[:8 * m])
    want_de, want_z = [[], []], [[], []]
    zc = [sl8(hx[0]), sl8(hx[1])]
    for k in range(depth):
        T = [{nm: sl8(trip[k][p][nm]) for nm in "abc"} for p in (0, 1)]
        ode = [ora.beaver_mask_mt(FID, zc[p], sl8(hy[p]), T[p]["a"], T[p]["b"]) for p in (0, 1)]
        nz = []
        for p in (0, 1):
            my_de, w = ora.batch_mul_9pass_mt(FID, p, keys[p], zc[p], sl8(hy[p]), T[p]["a"], T[p]["b"], T[p]["c"], ode[1 - p])
            want_de[p].append(my_de); want_z[p].append(w); nz.append(w)
        zc = nz

    def exact_split(zbuf, debuf):
        good = 0
        for k in range(depth):
            for p in (0, 1):
                o = zbuf[p][k].cpu().numpy().view(np.uint64)
                got = np.concatenate([o[:4 * m].reshape(m, 4), o[4 * n:4 * n + 4 * m].reshape(m, 4)], axis=1)
                d_ = debuf[p][k].cpu().numpy().view(np.uint64)
                g = (got == want_z[p][k].reshape(m, 8)).all(axis=1)
                g &= (d_[:4 * m].reshape(m, 4) == want_de[p][k][:4 * m].reshape(m, 4)).all(axis=1) & (d_[4 * n:4 * n + 4 * m].reshape(m, 4) == want_de[p][k][4 * m:].reshape(m, 4)).all(axis=1)
                good += int(g.sum())
        return good

    def wipe(bufs):
        for row in bufs:
            for t_ in row:
                t_.zero_()

    total = 2 * depth * m
    st0 = eng.stats()
    record("prefetched_async", "arkmpc_batch_from_host_async from pinned vectors (arkmpc_host_alloc: what a source that keeps its triples for this engine hands over): "
           "k_import_split reads the records in place over the link and writes the columns; gate k+1's imports are issued after gate k's K1", lambda: run_batches(pinned_src, True, True))
    st1 = eng.stats()
    ex_a = exact_split(z, de); wipe(z); wipe(de)
    record("async_no_prefetch", "the same imports issued only when the gate needs them (ARKMPC_TRIPLE_PREFETCH=0 in the host mirror)", lambda: run_batches(pinned_src, True, False))
    ex_b = exact_split(z, de); wipe(z); wipe(de)
    record("pageable_async", "the same from NEW pageable vectors for every gate (numpy / Vec memory): pinned in place by the import (hipHostRegister), DMA into a staging block, split kernel from HBM, unpinned at release",
           run_pageable)
    ex_c = exact_split(z, de); wipe(z); wipe(de)
    record("round4_blocking", "arkmpc_batch_from_host as it was: blocking copy on the compute stream into a staging block, then a split pass, three times per party-gate, "
           "nothing overlapped", lambda: run_batches(pinned_src, False, False))
    ex_d = exact_split(z, de)
    # the same three with a network round of NET_MS per gate between K1 and K2+K3: what reading ahead is for -- without it the link idles during
    # every round and the round idles during every upload
    NET_MS = 2.0
    with_net = {}
    for name, (asyn, pre) in (("prefetched_async", (True, True)), ("async_no_prefetch", (True, False)), ("round4_blocking", (False, False))):
        run_batches(pinned_src, asyn, pre, NET_MS)
        ts_ = [run_batches(pinned_src, asyn, pre, NET_MS) for _ in range(2)]
        with_net[name] = {"ms_per_gate": float(np.median(ts_)) * 1e3 / depth, "party_gates_per_s": 2 * n * depth / float(np.median(ts_))}
    ex_e = exact_split(z, de)
    async_imports = st1["batch_async_imports"] - st0["batch_async_imports"]
    # sessions with resident operands (AoS records)
    zs = [[torch.empty(8 * n, dtype=torch.int64, device="cuda") for _ in range(depth)] for _ in (0, 1)]
    des = [[torch.empty(8 * n, dtype=torch.int64, device="cuda") for _ in range(depth)] for _ in (0, 1)]
    record("sessions_resident_operands", "arkmpc_hostmul_begin_range / _finish_async with x, y, both payloads and the result RESIDENT (arkworks records in HBM) and a, b, c "
           "read in place from pinned host memory by the phase kernels: no import pass and no staging, but no read-ahead either", run_sessions)
    good = 0
    for k in range(depth):
        for p in (0, 1):
            o = zs[p][k].cpu().numpy().view(np.uint64)[:8 * m].reshape(m, 8)
            d_ = des[p][k].cpu().numpy().view(np.uint64)
            g = (o == want_z[p][k].reshape(m, 8)).all(axis=1)
            g &= (d_[:4 * m].reshape(m, 4) == want_de[p][k][:4 * m].reshape(m, 4)).all(axis=1) & (d_[4 * n:4 * n + 4 * m].reshape(m, 4) == want_de[p][k][4 * m:].reshape(m, 4)).all(axis=1)
            good += int(g.sum())
    ok = ex_a == total and ex_b == total and ex_c == total and ex_d == total and ex_e == total and good == total and async_imports == 4 * 6 * depth
    for q in hold:
        lib.arkmpc_host_free(q)
    best = modes["prefetched_async"]
    summary = {"circuit_party_gates_per_s": best["party_gates_per_s"], "circuit_two_party_gates_per_s": best["party_gates_per_s"] / 2,
               "circuit_frac_of_link_floor": best["frac_of_link_floor"]}
    return summary, {"what": "depth-%d chain z <- z * y of batch_mul gates at 2^%d, operands resident (split columns), both parties on this ONE GPU and its one host link, d||e handed "
                    "over in HBM, every gate on fresh random triples from host memory (192 B per party-gate over the link); wall clock from the first import to the last "
                    "kernel, the first gate's triples NOT read ahead" % (depth, log2n),
            "party_gates_per_s": best["party_gates_per_s"], "frac_of_link_floor": best["frac_of_link_floor"], "ms_per_gate_both_parties": best["ms_per_gate"],
            "link_floor": {"bytes_up_per_party_gate": 192, "measured_h2d_GBps": cal["h2d_GBps"], "floor_ms": floor_ms, "floor_party_gates_per_s": cal["h2d_GBps"] * 1e9 / 192,
                           "note": "both parties share this GPU's one link: per party-gate the floor is the same as for one party per GPU"},
            "modes": modes, "speedup_over_round4_path": modes["round4_blocking"]["ms"] / best["ms"],
            "with_network_round": {"net_round_ms_per_gate": NET_MS, "modes": with_net,
                                   "speedup_over_round4_path": with_net["round4_blocking"]["ms_per_gate"] / with_net["prefetched_async"]["ms_per_gate"],
                                   "what": "the same chain with a %.1f ms network round per gate between K1 and K2+K3 (host sleep after the payload is complete): read ahead, the "
                                           "next gate's triples cross the link during the round; otherwise link and network take turns" % NET_MS},
            "results_check": "every gate of the chain, both parties, d||e and result records == oracle (%s of each batch; %d party-gates x 5 runs), and every import of the "
                             "headline mode went up asynchronously (%d): %s" % ("ALL gates" if m == n else "the first 2^%d" % int(np.log2(m)), total, async_imports, "ok" if ok else "FAILED")}, ok

