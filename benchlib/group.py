"""Multi-GPU legs: the ordered all-gather of opened values over torch.distributed (N > 1 ranks), and the single-process modes that drive
N members through the C ABI's multi-device group (arkmpc_group_*), as a Rust party -- one process -- would."""
import ctypes
import importlib
import json
import os
import time

import numpy as np
import torch

from .common import (ALG_BYTES_K3, E2E_DOWN_BYTES, E2E_UP_BYTES, FID, HBM_PEAK_GBPS, METRIC, ROOT, build_workload, check_results, load_oracle,
                     oracle_sample, pcie_calibration, pinned_array, prepare_step, step)


def leg_gather(dist, world, rank, backend):
    """Ordered all-gather of the opened-value buffers in BASELINE config 5's shape: 2^24 / 8 = 2^21 scalars = 64 MiB per rank,
    straight into the final ordered buffer (sharding.gather_ordered, even shards -> all_gather_into_tensor, no pad / cat)."""
    sharding = importlib.import_module("ark-mpc_amd.sharding")
    per = 1 << 21
    dev = "cuda" if backend == "nccl" else "cpu"
    local = torch.full((4 * per,), rank + 1, dtype=torch.int64, device=dev)
    full = sharding.gather_ordered(local, per * world, 4)
    torch.cuda.synchronize(); dist.barrier()
    reps = 10
    t0 = time.perf_counter()
    for _ in range(reps):
        full = sharding.gather_ordered(local, per * world, 4)
    torch.cuda.synchronize(); dist.barrier()
    ms = (time.perf_counter() - t0) / reps * 1e3
    ok = all(int(full[4 * per * r].item()) == r + 1 and int(full[4 * per * (r + 1) - 1].item()) == r + 1 for r in range(world))
    return {"what": "ordered all-gather of opened values, 64 MiB per rank (config 5 shape)", "ms": ms, "bytes_per_rank": 32 * per,
            "bus_GBps_per_rank": 32 * per * (world - 1) / (ms * 1e-3) / 1e9, "ordered": ok}


def leg_group_end_to_end(pkg, devs, log2n, reps=6):
    """--single-process --only-e2e: the host-to-host path of leg_end_to_end for a party that owns SEVERAL GPUs and is ONE process
    (fabric.rs:402-466), through the group sessions of the C ABI (arkmpc_group_hostmul_*): one set of host record vectors of n = G * 2^log2n
    gates, member g running gates [g n/G, (g+1) n/G) over ITS device's PCIe link.  Host-fed a party is link-bound 20x below the kernels'
    rate, so the links are what more GPUs add; this leg prints every member's link rate and their sum.  Sessions of the two parties alternate
    (the peer's payload precomputed), vectors registered once by the caller, so both phases run as kernels on the vectors in place."""
    G = len(devs)
    per = 1 << log2n
    n = per * G
    torch.cuda.set_device(devs[0])
    eng = pkg.Engine(FID, device=devs[0], host_buffers=False, stream=torch.cuda.current_stream().cuda_stream)
    lib = pkg.load_library()
    parties, truth = build_workload(eng, n, seed=0xA11CE0E5, layout="aos")
    calls = prepare_step(eng, n, parties, "aos", chunks=max(1, n >> 20))
    step(calls)
    torch.cuda.synchronize()
    host = lambda t: np.ascontiguousarray(t.cpu().numpy().view(np.uint64))
    hold = []

    def pinned(arr):
        a_, q = pinned_array(lib, arr.size)
        a_[:] = arr
        hold.append(q)
        return a_

    H = [{k: pinned(host(getattr(p, k))) for k in "xyabc"} for p in parties]
    # (the device pipeline chunks d||e per 2^20 gates: rebuild the full d || e vectors)
    chunks = max(1, n >> 20)
    def full_de(t):
        v = host(t).reshape(chunks, 2, n // chunks, 4)
        return np.ascontiguousarray(np.concatenate([v[:, 0].reshape(-1), v[:, 1].reshape(-1)]))
    want_de = [pinned(full_de(p.de)) for p in parties]
    want_out = [host(p.out) for p in parties]
    keys = [p.key for p in parties]
    del parties, truth, calls
    torch.cuda.empty_cache()
    cal = pcie_calibration()
    de = [pinned(np.zeros(8 * n, dtype=np.uint64)) for _ in (0, 1)]
    out = [pinned(np.zeros(8 * n, dtype=np.uint64)) for _ in (0, 1)]
    grp = pkg.Group(FID, devs)

    def session(p, timers=None):
        if timers is not None:
            for m in range(G):
                lib.arkmpc_kernel_timer_arm(grp.member_ctx(m), ctypes.c_int(timers))
        s_ = grp.hostmul_begin(n, H[p]["x"], H[p]["y"], H[p]["a"], H[p]["b"], H[p]["c"], de[p])
        grp.hostmul_wait_de(s_)
        if timers is not None:
            for m in range(G):
                lib.arkmpc_kernel_timer_arm(grp.member_ctx(m), ctypes.c_int(timers + 1))
        grp.hostmul_finish(s_, p, keys[p], want_de[1 - p], out[p])

    session(0); session(1)
    ok = all(np.array_equal(de[p], want_de[p]) and np.array_equal(out[p], want_out[p]) for p in (0, 1))
    for p in (0, 1):
        de[p].fill(0); out[p].fill(0)
    ts = []
    t0 = time.perf_counter()
    for k in range(reps):
        t_ = time.perf_counter(); session(k & 1); ts.append(time.perf_counter() - t_)
    t = (time.perf_counter() - t0) / reps
    ok = ok and all(np.array_equal(de[p], want_de[p]) and np.array_equal(out[p], want_out[p]) for p in (0, 1))
    # one more session with the members' phase kernels timed (dispatch-bound HIP events on every member's context)
    session(0, timers=0)
    per_member = []
    ms = ctypes.c_float(0)
    single_launch = per <= (1 << 20)
    for m in range(G):
        lo, cnt = grp.shard_range(n, m)
        row = {"member": m, "device": devs[m], "gates": cnt, "path": grp.member_stats(m)["hostmul_zero_copy_phases"]}
        if single_launch:                                       # (a phase is one launch per 2^20 gates: the timer binds to the first)
            lib.arkmpc_kernel_timer_ms(grp.member_ctx(m), ctypes.c_int(0), ctypes.byref(ms)); p1 = ms.value
            lib.arkmpc_kernel_timer_ms(grp.member_ctx(m), ctypes.c_int(1), ctypes.byref(ms)); p2 = ms.value
            row.update({"phase1_kernel_ms": p1, "phase2_kernel_ms": p2,
                        "phase1_link_up_GBps": cnt * 256 / (p1 * 1e-3) / 1e9 if p1 > 0 else None,      # a, b, x, y records (only the share halves of x, y are used, but the link moves 64-byte reads)
                        "phase2_link_up_GBps": cnt * 128 / (p2 * 1e-3) / 1e9 if p2 > 0 else None})     # c records + the peer's d||e
        row["session_link_up_GBps"] = cnt * E2E_UP_BYTES / t / 1e9
        per_member.append(row)
    distinct = len(set(devs))
    ora = load_oracle()
    m_ = oracle_sample(n)
    sl8 = lambda a_: np.ascontiguousarray(a_[:8 * m_])
    ode = [ora.beaver_mask_mt(FID, sl8(H[p]["x"]), sl8(H[p]["y"]), sl8(H[p]["a"]), sl8(H[p]["b"])) for p in (0, 1)]
    exact = 0
    for p in (0, 1):
        my_de, w = ora.batch_mul_9pass_mt(FID, p, keys[p], sl8(H[p]["x"]), sl8(H[p]["y"]), sl8(H[p]["a"]), sl8(H[p]["b"]), sl8(H[p]["c"]), ode[1 - p])
        good = (out[p][:8 * m_].reshape(m_, 8) == w.reshape(m_, 8)).all(axis=1)
        good &= (de[p][:4 * m_].reshape(m_, 4) == my_de[:4 * m_].reshape(m_, 4)).all(axis=1) & (de[p][4 * n:4 * n + 4 * m_].reshape(m_, 4) == my_de[4 * m_:].reshape(m_, 4)).all(axis=1)
        exact += int(good.sum())
    ok = ok and exact == 2 * m_
    grp.close(); eng.close()
    for q in hold:
        lib.arkmpc_host_free(q)
    # the reference's own bench shape (benches/batch_ops.rs:19-39: share x, share y, batch_mul, open_authenticated_batch; both parties in-process, time =
    # max over the parties) for a party over this group, through the C++ host mirror (GroupFabric::batch_mul_host + the sharded opening)
    ref_shape = None
    exe = os.path.join(ROOT, "ark-mpc_amd", "lib", "arkmpc_host_bench")
    if os.path.exists(exe):
        import subprocess
        try:
            r_ = subprocess.run([exe, "group_batch_ops", str(n), "2"], capture_output=True, text=True, timeout=300,
                                env=dict(os.environ, ARKMPC_GROUP_DEVICES=",".join(str(d) for d in devs), ARKMPC_MOCK_LINK="host"))
            dd = json.loads(r_.stdout.strip().splitlines()[-1])
            ref_shape = {"ms": dd["seconds"] * 1e3, "elements_per_s": dd["elements_per_s"],
                         "what": "benches/batch_ops.rs:19-39 as written for n = %d over the group (host/bench_main.cpp group_batch_ops): batch_share_scalar x 2, batch_mul as a group "
                                 "session on host vectors, open_authenticated_batch on shards (two sequential SHA3-256 sponges over 32 n bytes per party: the floor of this shape)" % n}
        except Exception as ex:      # noqa: BLE001
            ref_shape = {"error": repr(ex)[:200]}
    res = {"what": "host arkworks records in -> host records out through ONE group session per batch_mul: n = %d x 2^%d gates per party, member g on gates [g n/G, (g+1) n/G) of the "
                   "same host vectors over its own device's link (arkmpc_group_hostmul_*); vectors pinned by the caller, sessions of the two parties alternating back to back"
                   % (G, log2n),
           "members": G, "devices": devs, "distinct_devices": distinct, "oversubscribed": distinct < G,
           "oversubscribed_note": ("the members share %d physical GPU(s) and therefore %d host link(s): the sum below is bounded by that, it is NOT an N-link measurement"
                                   % (distinct, distinct)) if distinct < G else None,
           "ms_per_session": t * 1e3, "ms_each_session": [round(x * 1e3, 3) for x in ts], "party_gates_per_s": n / t,
           "link_up_GBps_sum_over_members": n * E2E_UP_BYTES / t / 1e9, "link_down_GBps_sum_over_members": n * E2E_DOWN_BYTES / t / 1e9,
           "per_member": per_member, "measured_pcie_one_link": cal, "reference_bench_shape": ref_shape,
           "frac_of_links": (n * E2E_UP_BYTES / t / 1e9) / (cal["h2d_GBps"] * distinct),
           "results_check": "both parties' d||e and result records == the device-resident pipeline's on all %d gates, and == oracle on %s (%d of %d party-gates exact): %s"
                            % (n, "ALL of them" if m_ == n else "the first 2^%d" % int(np.log2(m_)), exact, 2 * m_, "ok" if ok else "FAILED")}
    return res, ok


def main_single_process(args):
    """N GPUs, ONE process: the multi-device group of the C ABI (include/arkmpc.h arkmpc_group_*).  Each party is a group over the same
    devices; member g of both parties lives on device g and owns gates [g*n/G, (g+1)*n/G) of a step of n = G * 2^log2n gates (weak
    scaling: 2^log2n gates per member).  Step = K1(P0), K1(P1), K2+K3(P0), K2+K3(P1) as four group calls; the d||e exchange is the
    member-by-member pointer hand-over (both parties' member g share device g).  Timing: barrier = group sync of both parties."""
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU fallback")
    pkg = importlib.import_module("ark-mpc_amd")
    devs = [int(d) for d in args.devices.split(",")] if args.devices else list(range(args.gpus))
    if len(devs) != args.gpus:
        raise SystemExit("--devices must list --gpus ids")
    G = len(devs)
    if args.only_e2e:
        r, ok = leg_group_end_to_end(pkg, devs, args.e2e_log2n)
        print(json.dumps(r), flush=True)
        if not ok:
            raise SystemExit("result check failed")
        return
    if args.log2n is None:
        args.log2n = 21 if G == 8 else 20
    per = 1 << args.log2n
    n = per * G
    layout = args.layout
    L = pkg.Group.SPLIT if layout == "split" else pkg.Group.AOS
    grp = [pkg.Group(FID, devs) for _ in (0, 1)]
    nsets = max(1, args.sets)
    # per member: the same seeded workload generator as the one-process-per-GPU path (seed + member = seed + rank)
    sets = []           # sets[k][member] = (parties, truth)
    engs = []
    for m, d in enumerate(devs):
        torch.cuda.set_device(d)
        engs.append(pkg.Engine(FID, device=d, host_buffers=False, stream=torch.cuda.current_stream().cuda_stream))
    for k in range(nsets):
        row = []
        for m, d in enumerate(devs):
            torch.cuda.set_device(d)
            ks = None if m == 0 else [row[0][0][pid].key for pid in (0, 1)]          # one party = one MAC key share, on every member
            row.append(build_workload(engs[m], per, seed=0xA11CE002 + m + 7919 * k, layout=layout, key_shares=ks))
        sets.append(row)
    for d in set(devs):
        torch.cuda.synchronize(d)
    calls = []          # calls[k] = [k1_p0, k1_p1, k3_p0, k3_p1]
    for k in range(nsets):
        sh = lambda pid, nm: [getattr(sets[k][m][0][pid], nm) for m in range(G)]
        k1k3 = [grp[pid].prepare_beaver(L, n, pid, sets[k][0][0][pid].key, sh(pid, "x"), sh(pid, "y"), sh(pid, "a"), sh(pid, "b"), sh(pid, "c"),
                                        sh(pid, "de"), sh(1 - pid, "de"), sh(pid, "out")) for pid in (0, 1)]
        calls.append([k1k3[0][0], k1k3[1][0], k1k3[0][1], k1k3[1][1]])
    lib = pkg.load_library()

    def barrier():
        grp[0].sync(); grp[1].sync()

    def step(k, arm_slot=None):
        # each party's group has its own member streams: the d||e hand-over is ordered on the device, member by member (arkmpc_group_wait_group).
        # Before the K1s: a party's K1 overwrites the d||e shards the PEER's previous K2+K3 read (write after read).  Before the K2+K3s: a
        # party's K2+K3 reads the d||e shards the peer's K1 writes (read after write).  Both waits of a pair are issued before either launch, so
        # the two parties' kernels of one phase stay free to overlap.
        for j, c in enumerate(calls[k]):
            if j in (0, 2):
                grp[0].wait_group(grp[1]); grp[1].wait_group(grp[0])
            if arm_slot is not None:            # dispatch-bound HIP events on every member's launch of this call
                g = grp[0 if j in (0, 2) else 1]
                for m in range(G):
                    lib.arkmpc_kernel_timer_arm(g.member_ctx(m), ctypes.c_int(arm_slot + j))
            c()

    def region(settle_ms, warmup, steps):
        barrier()
        if settle_ms > 0:
            t_s = time.perf_counter(); k = 0
            while (time.perf_counter() - t_s) * 1e3 < settle_ms:
                for _ in range(8):
                    step(k % nsets); k += 1
                barrier()
        for w in range(warmup):
            step(w % nsets)
        barrier()
        sampled = [s_ for s_ in range(steps) if s_ % max(1, steps // 8) == 0][:8]
        slot_of = {s_: 4 * i for i, s_ in enumerate(sampled)}
        t0 = time.perf_counter()
        for s_ in range(steps):
            step(s_ % nsets, slot_of.get(s_))
        barrier()
        elapsed = time.perf_counter() - t0
        per_member = []
        for m in range(G):
            ms = ctypes.c_float(0)
            acc = [0.0, 0.0]
            for s_ in sampled:
                for j in range(4):
                    g = grp[0 if j in (0, 2) else 1]
                    lib.arkmpc_kernel_timer_ms(g.member_ctx(m), ctypes.c_int(slot_of[s_] + j), ctypes.byref(ms))
                    acc[0 if j < 2 else 1] += ms.value
            cnt = max(1, 2 * len(sampled))
            per_member.append({"member": m, "device": devs[m], "k1_avg_launch_ms": acc[0] / cnt, "k3_avg_launch_ms": acc[1] / cnt,
                               "kernel_ms_per_step": (acc[0] + acc[1]) / max(1, len(sampled))})
        return elapsed, per_member

    cold = None
    if not args.no_cold and args.settle_ms > 0:
        cold = region(0, args.warmup, args.steps)
    elapsed, per_member = region(args.settle_ms, args.warmup, args.steps)
    # results: every member's range opens to x*y with a valid MAC (engine ops on that member's device)
    ok = True
    if not args.no_check:
        for k in range(min(nsets, args.steps)):
            for m, d in enumerate(devs):
                torch.cuda.set_device(d)
                ps, tr = sets[k][m]
                ok = ok and check_results(engs[m], per, ps, tr, layout)
    # gather of opened values in config 5's shape: 2^21 scalars (64 MiB) per member into one ordered buffer on member 0 / on every member
    gather = None
    if not args.no_gather:
        gper = 1 << 21
        gn = gper * G
        sh = grp[0].malloc(gn, 1, 4)
        root_buf = torch.empty(4 * gn, dtype=torch.int64, device="cuda:%d" % devs[0])
        outs = [torch.empty(4 * gn, dtype=torch.int64, device="cuda:%d" % d) for d in devs]
        host = np.arange(4 * gn, dtype=np.uint64)
        grp[0].scatter_h2d(gn, 1, 4, host, sh)
        res = {}
        for name, fn in (("gather_to_member0", lambda: grp[0].gather(gn, 1, 4, sh, 0, root_buf)), ("allgather", lambda: grp[0].allgather(gn, 1, 4, sh, outs))):
            fn(); grp[0].sync()
            reps = 10
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            grp[0].sync()
            ms = (time.perf_counter() - t0) / reps * 1e3
            moved = 32 * gper * (G - 1) * (1 if name.startswith("gather") else G)
            res[name] = {"ms": ms, "bytes_moved_between_members": moved, "GBps": moved / (ms * 1e-3) / 1e9 if ms > 0 else None}
        ordered = bool(np.array_equal(root_buf.cpu().numpy().view(np.uint64), host)) and all(bool(np.array_equal(o.cpu().numpy().view(np.uint64), host)) for o in outs)
        res["what"] = "ordered gather of opened values, 64 MiB per member (config 5 shape), as direct peer writes (hipMemcpyPeerAsync pushes on the source's stream)"
        res["ordered"] = ordered
        res["peer_access_all_pairs"] = all(grp[0].peer_access(a, b) for a in range(G) for b in range(G))
        ok = ok and ordered
        grp[0].free(sh)
        gather = res
    gates = n * args.steps
    k3_ms = float(np.mean([pm["k3_avg_launch_ms"] for pm in per_member]))
    k1_ms = float(np.mean([pm["k1_avg_launch_ms"] for pm in per_member]))
    ach = per * ALG_BYTES_K3 / (k3_ms * 1e-3) / 1e9
    distinct = len(set(devs))
    out = {
        "metric": METRIC, "value": gates / elapsed, "unit": "gates/s", "n_gpus": G, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u256 Montgomery (8 x u32 limbs, v_mad_u64_u32)", "data": "synthetic",
        "mode": "single-process: ONE process drives all members through arkmpc_group_* (include/arkmpc.h)",
        "ordering": "the two parties' groups are ordered member by member with arkmpc_group_wait_group before every K1 pair (write after read) and every K2+K3 pair (read after write)",
        "ranks_seen": G, "devices": devs, "distinct_devices": distinct,
        "oversubscribed": distinct < G,
        "per_member": per_member,
        "config": {"workload": "2^%d AuthenticatedScalar Beaver muls over BN254 Fr per member per step (%d members = %d gates per step), two parties in-process, "
                               "d||e handed over member by member (BASELINE.json configs[%d] shape)" % (args.log2n, G, n, 2 if (G == 8 and args.log2n == 21) else 1),
                   "gates_per_gpu": per, "gates_per_step_all_gpus": n, "field": "bn254_fr", "layout": layout, "launches_per_step": 4 * G,
                   "gates_per_launch": per, "workload_sets_rotated": nsets, "settle_ms": args.settle_ms,
                   "parallelism": "gate-range sharding inside the C ABI, no data-path collective"},
        "roofline": {"bound": "hbm", "kernel": "k_beaver_finish_asm (K2+K3 fused, hand-scheduled), mean over members", "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": ach / HBM_PEAK_GBPS, "traffic": None, "algorithmic_bytes_per_launch": per * ALG_BYTES_K3, "gates_per_launch": per, "avg_launch_ms": k3_ms,
                     "note": "per-member figure: with members sharing a GPU the launches of different members overlap and each one's duration stretches accordingly"},
        "pipeline": {"k1_avg_launch_ms": k1_ms, "k3_avg_launch_ms": k3_ms},
        "results_check": "open(batch_mul(x,y)) == x*y and MAC shares sum to key*x*y on every member's range: %s" % ("ok" if ok else "FAILED"),
    }
    if cold is not None:
        out["value_cold"] = gates / cold[0]
    if gather is not None:
        out["gather"] = gather
    print(json.dumps(out), flush=True)
    for g in grp:
        g.close()
    for e in engs:
        e.close()
    if not ok:
        raise SystemExit("result check failed")

