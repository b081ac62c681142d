"""End-to-end leg (SURVEY 8(d) "an end-to-end figure including H2D/D2H"): host arkworks records in, host records out, through the
streaming sessions of the C ABI.  Never the metric's `value`, which is quoted with inputs resident in HBM."""
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

from .common import (E2E_DOWN_BYTES, E2E_UP_BYTES, FID, ROOT, build_workload, load_oracle, oracle_sample, pcie_calibration, prepare_step, step)


def leg_end_to_end(pkg, eng, dev, log2n=20, reps_min=6):
    """SURVEY 8(d) "an end-to-end figure including H2D/D2H", in the shape of the reference's own bench (benches/batch_ops.rs:19-39: host values
    in, host values out, both parties in-process, time = max over the parties): every operand starts as arkworks ScalarShare records in HOST
    memory (what a Rust Vec<ScalarShare> is), the d||e payloads cross the host link in both directions as they would on a real network, the
    result records end in host memory.  Runs the streaming sessions of the C ABI (arkmpc_hostmul_*: three-stream pipeline, buffers pinned in
    place).  one_party = what one party's process sees on its own GPU; two_party = both parties sharing THIS GPU and its one PCIe link."""
    import threading
    lib = pkg.load_library()
    n = 1 << log2n
    parties, truth = build_workload(eng, n, seed=0xA11CE0E2, layout="aos")
    calls = prepare_step(eng, n, parties, "aos")
    step(calls)
    torch.cuda.synchronize()
    host = lambda t: np.ascontiguousarray(t.cpu().numpy().view(np.uint64))
    H = [{k: host(getattr(p, k)) for k in "xyabc"} for p in parties]
    want_de = [host(p.de) for p in parties]            # the device-resident pipeline's buffers for the same workload (itself checked against the oracle below)
    want_out = [host(p.out) for p in parties]
    keys = [p.key for p in parties]
    del parties, truth, calls
    torch.cuda.empty_cache()
    cal = pcie_calibration()
    de = [np.empty(8 * n, dtype=np.uint64) for _ in (0, 1)]
    out = [np.empty(8 * n, dtype=np.uint64) for _ in (0, 1)]
    for a in de + out:
        a.fill(0)                                      # touched, like a Vec the caller has initialised
    ok = True

    def one_party(p, peer_de):
        s = eng.hostmul_begin(n, H[p]["x"], H[p]["y"], H[p]["a"], H[p]["b"], H[p]["c"], de[p])
        eng.hostmul_finish(s, p, keys[p], peer_de, out[p])

    def zero_copy_phases():
        return eng.stats()["hostmul_zero_copy_phases"]          # arkmpc_ctx_get_stats: which path ran (a count, nothing timed depends on it)

    def timed_one(label, fresh):
        """back_to_back: `reps` sessions one after the other, as a circuit of many gates keeps the link busy (the throughput figure).  isolated: one
        session after the link has idled for a few ms.  fresh = every session gets NEWLY ALLOCATED vectors (inputs copied, outputs zeroed,
        before the clock starts): a caller whose Vecs are new for every gate, the worst case for pinning in place.  Consecutive sessions
        alternate between the two parties' inputs: the device block is recycled from session to session, so a kernel that ran ahead of its upload
        would compute on the OTHER party's stale records and the check would catch it."""
        nonlocal ok
        one_party(0, want_de[1])                       # warm: device block, streams, events
        t_ = time.perf_counter(); one_party(0, want_de[1]); est = time.perf_counter() - t_
        reps = min(32, max(reps_min, int(np.ceil(0.04 / max(est, 1e-5)))))       # small batches: enough sessions for ~40 ms, so that one slow pin does not decide the mean

        def vectors(k):
            p = k & 1
            if not fresh:
                de[p].fill(0); out[p].fill(0)
                return p, H[p], de[p], out[p], want_de[1 - p]
            d_, o_ = np.empty(8 * n, dtype=np.uint64), np.empty(8 * n, dtype=np.uint64)
            d_.fill(0); o_.fill(0)
            return p, {k_: v.copy() for k_, v in H[p].items()}, d_, o_, want_de[1 - p].copy()

        def run(v):
            p, ins, d_, o_, peer = v
            s = eng.hostmul_begin(n, ins["x"], ins["y"], ins["a"], ins["b"], ins["c"], d_)
            eng.hostmul_finish(s, p, keys[p], peer, o_)

        def good(v):
            p, _, d_, o_, _ = v
            return bool(np.array_equal(d_, want_de[p]) and np.array_equal(o_, want_out[p]))

        sets_ = [vectors(k) for k in range(reps)]
        zc0 = zero_copy_phases()
        each = []
        t0 = time.perf_counter()
        for v in sets_:
            t_ = time.perf_counter(); run(v); each.append((time.perf_counter() - t_) * 1e3)
        t = (time.perf_counter() - t0) / reps
        zc1 = zero_copy_phases()
        ok = ok and all(good(v) for v in (sets_ if fresh else sets_[-2:]))
        del sets_
        iso = []
        for k in range(4):
            v = vectors(k)
            time.sleep(0.004)
            t1 = time.perf_counter()
            run(v)
            iso.append(time.perf_counter() - t1)
            ok = ok and good(v)
        zc = [(b - a) / reps for a, b in zip(zc0, zc1)]
        return {"buffers": label, "path": {"phase1": "zero-copy kernel on the caller's vectors" if zc[0] == 1 else "copy pipeline" if zc[0] == 0 else "mixed",
                                            "phase2": "zero-copy kernel on the caller's vectors" if zc[1] == 1 else "copy pipeline" if zc[1] == 0 else "mixed"},
                "ms": t * 1e3, "sessions_timed": reps, "ms_median_session": float(np.median(each)), "ms_each_session": [round(x_, 3) for x_ in each[:12]], "ms_isolated_call": float(np.median(iso)) * 1e3, "party_gates_per_s": n / t,
                "party_gates_per_s_isolated_call": n / float(np.median(iso)), "h2d_GBps": n * E2E_UP_BYTES / t / 1e9,
                "d2h_GBps": n * E2E_DOWN_BYTES / t / 1e9, "frac_of_measured_pcie": (n * E2E_UP_BYTES / t / 1e9) / cal["h2d_GBps"]}

    # registered FIRST: a kernel addresses a caller-registered vector in place only in the FIRST registered life of its addresses (the library retires
    # the addresses of every registration that ended, its own per-call ones included: csrc/arkmpc_internal.hpp PinRegistry), so these vectors are
    # registered before any pageable session has pinned and released them
    regs = [a for p in (0, 1) for a in list(H[p].values())] + de + out + want_de
    for a in regs:
        lib.arkmpc_host_register(ctypes.c_void_p(a.ctypes.data), ctypes.c_size_t(a.nbytes))
    registered = timed_one("registered once by the caller (arkmpc_host_register), as a caller that keeps its vectors across gates would: both phases run as kernels that read and write "
                           "the pinned vectors in place, no copy commands (ARKMPC_HOSTMUL_ZEROCOPY=0 puts them back on the copy pipeline: 8.0-8.1 ms at 2^20)", False)
    pageable = timed_one("pageable, NEW vectors for every session (numpy / Vec memory); pinned in place inside each call and moved by DMA (no kernel addresses a vector the library registered itself, DESIGN section 4)", True)
    # two parties on this one GPU, a context and a host thread each, payloads handed over in host memory (network/mock.rs moves host payloads)
    es = [pkg.Engine(FID, device=dev) for _ in (0, 1)]
    bar = threading.Barrier(2)
    spans = [[], []]
    marks = [[], []]
    errs = []

    def party(p, rounds):
        try:
            torch.cuda.set_device(dev)
            for _ in range(rounds):
                bar.wait(timeout=120)
                t0 = time.perf_counter()
                s = es[p].hostmul_begin(n, H[p]["x"], H[p]["y"], H[p]["a"], H[p]["b"], H[p]["c"], de[p])
                t1 = time.perf_counter()
                es[p].hostmul_wait_de(s)
                t2 = time.perf_counter()
                bar.wait(timeout=120)                  # the "network": the peer's payload is complete in host memory
                t3 = time.perf_counter()
                es[p].hostmul_finish(s, p, keys[p], de[1 - p], out[p])
                t4 = time.perf_counter()
                spans[p].append(t4 - t0)
                marks[p].append([(t1 - t0) * 1e3, (t2 - t0) * 1e3, (t3 - t0) * 1e3, (t4 - t0) * 1e3])
        except Exception as ex:      # noqa: BLE001
            errs.append(repr(ex))
            bar.abort()

    reps = reps_min
    th = [threading.Thread(target=party, args=(p, reps + 1)) for p in (0, 1)]
    for t in th: t.start()
    for t in th: t.join()
    two = None
    if errs:
        ok = False
        two = {"error": errs[:2]}
    else:
        per_round = [max(a, b) for a, b in zip(spans[0][1:], spans[1][1:])]      # round 0 = warm-up; time of a round = max over the parties
        t = float(np.median(per_round))
        ok = ok and all(np.array_equal(de[p], want_de[p]) and np.array_equal(out[p], want_out[p]) for p in (0, 1))
        two = {"ms": t * 1e3, "two_party_gates_per_s": n / t, "h2d_GBps": 2 * n * E2E_UP_BYTES / t / 1e9, "d2h_GBps": 2 * n * E2E_DOWN_BYTES / t / 1e9,
               "frac_of_measured_pcie": (2 * n * E2E_UP_BYTES / t / 1e9) / cal["h2d_GBps"],
               "marks_ms_last_round": {"what": "per party: begin returned, own d||e complete in host memory, peer's payload available, finish returned", "p0": marks[0][-1], "p1": marks[1][-1]},
               "what": "the same with one host thread + context PER PARTY (execute_mock_mpc's shape): the two parties' uploads race each other on the link and both lose"}
    for e_ in es:
        e_.close()
    # the same two parties driven by ONE host thread on ONE context, both sessions open at once (how an in-process mock -- one process, both
    # parties -- naturally drives one GPU): the two parties' uploads then queue on one stream instead of racing each other on the link
    two_threads = two
    eng1 = pkg.Engine(FID, device=dev)

    def one_thread_round():
        ss = [eng1.hostmul_begin(n, H[p]["x"], H[p]["y"], H[p]["a"], H[p]["b"], H[p]["c"], de[p]) for p in (0, 1)]
        for p in (0, 1):
            eng1.hostmul_wait_de(ss[p])
        for p in (0, 1):
            eng1.hostmul_finish(ss[p], p, keys[p], de[1 - p], out[p])

    one_thread_round()
    for p in (0, 1):
        de[p].fill(0); out[p].fill(0)
    ts1 = []
    for _ in range(reps):
        t0 = time.perf_counter(); one_thread_round(); ts1.append(time.perf_counter() - t0)
    ok1 = all(np.array_equal(de[p], want_de[p]) and np.array_equal(out[p], want_out[p]) for p in (0, 1))
    ok = ok and ok1
    eng1.close()
    t1 = float(np.median(ts1))
    two = {"ms": t1 * 1e3, "two_party_gates_per_s": n / t1, "h2d_GBps": 2 * n * E2E_UP_BYTES / t1 / 1e9, "d2h_GBps": 2 * n * E2E_DOWN_BYTES / t1 / 1e9,
           "frac_of_measured_pcie": (2 * n * E2E_UP_BYTES / t1 / 1e9) / cal["h2d_GBps"],
           "what": "both parties on this ONE GPU and its one PCIe link, ONE host thread and context driving both parties' sessions (begin, begin, wait, wait, finish, finish), "
                   "d||e handed over in host memory; 768 B up per two-party gate, so the link's floor is %.1f ms" % (2 * n * E2E_UP_BYTES / cal["h2d_GBps"] / 1e6),
           "two_host_threads_two_contexts": two_threads}
    # the same sessions with the payloads in their WIRE form (the frames QuicTwoPartyNet puts on the stream: serde_json text, ~115 bytes per scalar):
    # host records in -> frame out; peer's frame in -> host records out.  The text is rendered and parsed on the GPU; it crosses the link instead of
    # the raw scalars (about 3.6x their bytes each way).
    wire = None
    fbuf = []
    try:
        cap = eng.wire_frame_bound(2 * n)
        for _ in range(3):
            q = ctypes.c_void_p()
            if lib.arkmpc_host_alloc(ctypes.c_size_t(cap), ctypes.byref(q)) != 0:
                raise RuntimeError("arkmpc_host_alloc(frame)")
            fbuf.append((q, np.ctypeslib.as_array(ctypes.cast(q, ctypes.POINTER(ctypes.c_uint8)), shape=(cap,))))
        peer_frames, peer_lens = [], []
        for p in (0, 1):                                   # each party's own frame once, kept as the other party's inbound message
            s_, ln = eng.hostmul_begin_wire(n, H[p]["x"], H[p]["y"], H[p]["a"], H[p]["b"], H[p]["c"], 1000 + p, fbuf[2][1])
            eng.hostmul_abort(s_)
            peer_frames.append(fbuf[2][1][:ln].copy()); peer_lens.append(ln)
        for a in peer_frames:
            lib.arkmpc_host_register(ctypes.c_void_p(a.ctypes.data), ctypes.c_size_t(a.nbytes))
    except Exception as ex:      # noqa: BLE001
        wire = {"error": repr(ex)[:200]}
    if wire is None:
        try:
            def wire_session(k):
                p = k & 1
                s_, ln = eng.hostmul_begin_wire(n, H[p]["x"], H[p]["y"], H[p]["a"], H[p]["b"], H[p]["c"], 1000 + p, fbuf[p][1])
                rid = eng.hostmul_finish_wire(s_, p, keys[p], peer_frames[1 - p], peer_lens[1 - p], out[p])
                return p, ln, rid
            wire_session(0)
            tw = []
            okw = True
            for k in range(reps):
                out[k & 1].fill(0); fbuf[k & 1][1][:4096].fill(0)
                t0 = time.perf_counter(); p, ln, rid = wire_session(k); tw.append(time.perf_counter() - t0)
                okw = okw and ln == peer_lens[p] and rid == 1000 + (1 - p) and bool(np.array_equal(fbuf[p][1][:ln], peer_frames[p])) and bool(np.array_equal(out[p], want_out[p]))
            # the frame text itself against the serde_json model, on its head (header + the first 512 scalars of d) and its tail
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import pyref
            from helpers import from_mont_array
            for p in (0, 1):
                head_vals = from_mont_array(FID, want_de[p][:4 * 512])
                tail_vals = from_mont_array(FID, want_de[p][-4 * 512:])
                model_h = pyref.wire_frame("ScalarBatch", 1000 + p, pyref.wire_scalar_records(FID, head_vals))[8:-3]
                model_t = pyref.wire_frame("ScalarBatch", 0, pyref.wire_scalar_records(FID, tail_vals))[8:]
                fr = peer_frames[p].tobytes()
                t_text = model_t[model_t.index(b"[[") + 1:]
                okw = okw and fr[8:8 + len(model_h)] == model_h and fr.endswith(t_text) and int.from_bytes(fr[:8], "little") == len(fr) - 8
            tm = float(np.median(tw))
            wire = {"ms": tm * 1e3, "party_gates_per_s": n / tm, "frame_bytes": int(peer_lens[0]), "text_bytes_per_scalar": peer_lens[0] / (2.0 * n),
                    "link_bytes_per_party_gate": {"up": 320 + peer_lens[0] / n, "down": 64 + peer_lens[0] / n},
                    "what": "arkmpc_hostmul_begin_wire + _finish_wire, one party, pinned vectors and frame buffers, sessions back to back: records in, this party's "
                            "NetworkOutbound{ScalarBatch(d||e)} frame out; the peer's frame in, result records out.  The frames are rendered / validated and parsed on the GPU "
                            "(csrc/arkmpc_wire.hip); not overlapped with the phases (a frame's length is data dependent)",
                    "check": "frames == the serde_json model on head and tail and identical from session to session, result_id round-trips, results == the plain sessions': %s" % ("ok" if okw else "FAILED")}
            ok = ok and okw
        except Exception as ex:      # noqa: BLE001
            wire = {"error": repr(ex)[:300]}
            ok = False
        for a in peer_frames:
            lib.arkmpc_host_unregister(ctypes.c_void_p(a.ctypes.data))
    for a in regs:
        lib.arkmpc_host_unregister(ctypes.c_void_p(a.ctypes.data))
    for q, _ in fbuf:
        lib.arkmpc_host_free(q)
    del fbuf
    # the oracle on a sample of the same host data (the device-resident buffers used as the expectation above are not an independent witness)
    ora = load_oracle()
    m = oracle_sample(n)
    sl8 = lambda a: np.ascontiguousarray(a[:8 * m])
    ode = [ora.beaver_mask_mt(FID, sl8(H[p]["x"]), sl8(H[p]["y"]), sl8(H[p]["a"]), sl8(H[p]["b"])) for p in (0, 1)]
    exact = 0
    for p in (0, 1):
        my_de, w = ora.batch_mul_9pass_mt(FID, p, keys[p], sl8(H[p]["x"]), sl8(H[p]["y"]), sl8(H[p]["a"]), sl8(H[p]["b"]), sl8(H[p]["c"]), ode[1 - p])
        good = (out[p][:8 * m].reshape(m, 8) == w.reshape(m, 8)).all(axis=1)
        good &= (de[p][:4 * m].reshape(m, 4) == my_de[:4 * m].reshape(m, 4)).all(axis=1) & (de[p][4 * n:4 * n + 4 * m].reshape(m, 4) == my_de[4 * m:].reshape(m, 4)).all(axis=1)
        exact += int(good.sum())
    ok = ok and exact == 2 * m
    best = registered if registered["party_gates_per_s"] >= pageable["party_gates_per_s"] else pageable
    # the reference's own bench shape, literally (benches/batch_ops.rs:19-39: share x, share y, batch_mul, open_authenticated_batch; both parties
    # in-process, time = max over the parties), through the C++ host mirror: its own binary, run as a subprocess
    ref_shape = {}
    exe = os.path.join(ROOT, "ark-mpc_amd", "lib", "arkmpc_host_bench")
    if os.path.exists(exe):
        import subprocess
        for link in ("host", "device"):
            try:
                r = subprocess.run([exe, "batch_ops", str(n), "2"], capture_output=True, text=True, timeout=180, env=dict(os.environ, ARKMPC_MOCK_LINK=link))
                dd = json.loads(r.stdout.strip().splitlines()[-1])
                ref_shape[link + "_link"] = {"ms": dd["seconds"] * 1e3, "elements_per_s": dd["elements_per_s"]}
            except Exception as ex:      # noqa: BLE001
                ref_shape[link + "_link"] = {"error": repr(ex)[:200]}
        ref_shape["what"] = ("benches/batch_ops.rs:19-39 as written, n = 2^%d: batch_share_scalar x 2, batch_mul, open_authenticated_batch (two sequential SHA3-256 sponges over 32 n "
                             "bytes per party: ~45 ms each at 2^20, the floor of this shape), dummy Beaver source, host mirror (host/bench_main.cpp); host_link = payloads cross "
                             "as host vectors, device_link = as HBM buffers" % log2n)
    summary = {"end_to_end_party_gates_per_s": best["party_gates_per_s"], "end_to_end_two_party_gates_per_s": two.get("two_party_gates_per_s") if two else None,
               "end_to_end_frac_of_measured_pcie": best["frac_of_measured_pcie"]}
    return summary, {"what": "host arkworks records in -> host records out, 2^%d Beaver muls over BN254 Fr per party (benches/batch_ops.rs shape); NOT the metric's `value`, "
                    "which is quoted with inputs resident in HBM" % log2n,
            "bytes_per_party_gate": {"up": E2E_UP_BYTES, "down": E2E_DOWN_BYTES},
            "party_gates_per_s": best["party_gates_per_s"], "two_party_gates_per_s": two.get("two_party_gates_per_s") if two else None,
            "h2d_GBps": best["h2d_GBps"], "d2h_GBps": best["d2h_GBps"], "frac_of_measured_pcie": best["frac_of_measured_pcie"],
            "one_party": {"pageable": pageable, "registered": registered}, "two_party_one_gpu": two, "measured_pcie": cal,
            "wire_form": wire, "reference_bench_shape": ref_shape,
            "link_floor_note": "one PCIe gen5 x16 link: a party-gate needs 384 B up, so the link's measured %.1f GB/s allows at most %.3g party-gates/s "
                               "(and half of that per two-party gate when both parties share the link)" % (cal["h2d_GBps"], cal["h2d_GBps"] * 1e9 / E2E_UP_BYTES),
            "results_check": "all 2^%d gates of both parties == the device-resident pipeline's records, and %s == oracle, every word of d||e and result (%d of %d party-gates exact): %s"
                             % (log2n, "ALL of them" if m == n else "the first 2^%d gates" % int(np.log2(m)), exact, 2 * m, "ok" if ok else "FAILED")}, ok

