#!/usr/bin/env python3
"""bench.py -- authenticated Beaver mul-gates/sec over BN254 Fr, batch 2^20 per GPU (BASELINE.json config[1]).

A "step" is one complete two-party `AuthenticatedScalarResult::batch_mul` (reference:
online-phase/src/algebra/scalar/authenticated_scalar.rs:848-879) over a batch of n gates, both
parties simulated on the same GPU with the network mocked as a pointer swap (as
online-phase/benches/batch_ops.rs:20-39 runs both parties in-process):

    party0.K1 beaver_mask -> party1.K1 beaver_mask -> [exchange d||e] ->
    party0.K2+K3 beaver_finish_fused -> party1.K2+K3 beaver_finish_fused        (4 kernel launches)

All inputs are resident in HBM before the timed region.  Data is synthetic: uniformly random field
elements, valid Beaver triples and valid SPDZ MACs generated on the GPU from a fixed seed.

Launch:  python bench.py [--gpus N --steps K --warmup W]         (N = 1)
         python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
Rank 0 prints ONE JSON line.  Multi-GPU = the same batch size on every rank (weak scaling), gates
are independent so there is no data-path collective; ranks only meet at the timing barriers.
"""
import argparse
import ctypes
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

METRIC = "authenticated Beaver mul-gates/sec over BN254 Fr, batch 2^20, at 1/2/4/8 GPUs"
HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
ALG_BYTES_PER_GATE = 1024       # two-party gate, SURVEY.md section 8(d)
# apportioning of the 512 B / party-gate of SURVEY 8(d) between the two kernels of a party:
ALG_BYTES_K1 = 192              # read x,y shares+MACs 128, write own d||e 64
ALG_BYTES_K3 = 320              # read a,b,c shares+MACs 192, read peer d||e 64, write result 64
FID = 0                         # BN254 Fr


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--log2n", type=int, default=None, help="gates per GPU per step (default 2^20, the metric's batch; with 8 ranks 2^21 = "
                    "BASELINE config 3's 2^24 gates over 8 GPUs; steps above 2^20 gates run as 2^20-gate ranges, so the per-gate work is identical)")
    ap.add_argument("--layout", choices=["aos", "split"], default="split",
                    help="HBM layout of share vectors: arkworks AoS (drop-in) or engine-native split columns")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-log2n", type=int, default=20, help="CPU baseline sample size (gates)")
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--settle-ms", type=float, default=30.0, help="untimed: run the pipeline for this long BEFORE the W warm-up steps, on every rank, so that the "
                    "timed region does not sit inside the power controller's transient after idle (probes/ramp_probe.py: a cold MI355X runs the first steps fast, "
                    "then creeps from 0.195 to 0.21-0.25 ms per step for several ms before settling at 0.198).  0 disables.  Reported in config.settle_ms.")
    ap.add_argument("--single-process", action="store_true", help="N GPUs driven by ONE process through the C ABI's multi-device group (arkmpc_group_*): what a "
                    "Rust party, which is one process, would run.  The driver's N>1 runs use torch.distributed.run (one process per GPU); this mode is the same "
                    "sharding behind the FFI.  Reports ranks_seen, per-member kernel times and the peer-write gather rate.")
    ap.add_argument("--devices", default=None, help="--single-process: comma-separated device ids of the members (default 0..N-1); ids may repeat "
                    "(members then share a GPU: how the mode is exercised on a one-GPU box)")
    ap.add_argument("--no-cold", action="store_true", help="skip the cold pass (same W + K region with --settle-ms 0, run first) reported as value_cold / frac_cold")
    ap.add_argument("--no-extras", action="store_true", help="skip the extra legs reported next to the headline at N=1 (AoS layout, config 4, config 5)")
    ap.add_argument("--min-timed-ms", type=float, default=50.0, help="the timed region repeats the K steps in whole rounds until it is at least this long "
                    "(K = 20 steps of 0.19 ms would be a 3.8 ms region: too short for the driver's clock and the power controller); 0 = exactly K steps. "
                    "Reported: steps = K, timed_rounds, timed_steps_total, timed_region_ms; ms_per_step and value are over the whole region")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak", help="weak: --log2n gates per GPU per step whatever N (the default, what the driver runs); "
                    "strong: 2^--total-log2n gates per step in total (BASELINE config 3: 2^24), cut into N contiguous ranges")
    ap.add_argument("--total-log2n", type=int, default=24, help="--scaling strong: total gates per step over all GPUs")
    ap.add_argument("--only-e2e", action="store_true", help="run only the end-to-end (host records in, host records out) leg and print its JSON")
    ap.add_argument("--e2e-log2n", type=int, default=20, help="gates per party of the end-to-end leg")
    ap.add_argument("--only-circuit", action="store_true", help="run only the circuit leg (resident operands, triples from host memory) and print its JSON")
    ap.add_argument("--circuit-log2n", type=int, default=20, help="gates per batch_mul of the circuit leg")
    ap.add_argument("--circuit-depth", type=int, default=8, help="dependent gates in the circuit leg's chain")
    ap.add_argument("--no-gather", action="store_true", help="N>1: skip the timed ordered all-gather of opened-value buffers (config 5 shape, 64 MiB per rank)")
    ap.add_argument("--dist-backend", default="nccl", help="torch.distributed backend for N>1 (nccl = RCCL; gloo only for launch-path tests)")
    ap.add_argument("--sets", type=int, default=2, help="independent workload sets rotated step by step, so that no input line of step s "
                    "can still be cached (256 MiB Infinity Cache) when step s+1 runs; 1 = reuse the same buffers every step")
    ap.add_argument("--event-every", type=int, default=4, help="time the four kernels of every k-th timed step with dispatch-bound HIP events (at most 16 steps); "
                    "the whole region is bracketed by one event pair regardless")
    ap.add_argument("--k3-order", default="01", choices=["01", "10"], help="order of the two parties' K2+K3 launches after K1(P0), K1(P1). "
                    "The parties are independent; measured: no difference (within +-1 %).")
    ap.add_argument("--chunks", type=int, default=0, help="split each step's batch into this many gate ranges, each run K1,K1,K3,K3. "
                    "0 = automatic: ranges of 2^20 gates, so that the d||e buffers both parties exchange (128 MiB per range) stay in "
                    "the 256 MiB Infinity Cache between K1 and K3 -- measured: 2^21 gates/step 4.6e9 -> 5.1e9 gates/s, 2^22: 4.7e9 -> 5.2e9; "
                    "smaller ranges lose (2^20 in two halves: 4.6e9)")
    return ap.parse_args()


def rand_field_elems(eng, n, gen):
    """n uniformly random BN254 Fr elements in Montgomery form (int64 tensor of 4n limbs), generated on the GPU."""
    raw = torch.randint(-(2**63), 2**63 - 1, (4 * n,), dtype=torch.int64, device="cuda", generator=gen)
    out = torch.empty_like(raw)
    eng.scalar_from_canonical(n, raw, out)   # reduces mod p, then to Montgomery form
    return out


def additive_split(eng, n, v, gen):
    s0 = rand_field_elems(eng, n, gen)
    s1 = torch.empty_like(s0)
    eng.scalar_sub(n, v, s0, s1)
    return s0, s1


def make_shares(eng, n, v, key, gen, layout):
    """SPDZ-share the vector v under MAC key `key` (both Montgomery limb tensors) -> per-party share buffers."""
    mac = torch.empty_like(v)
    eng.scalar_mul(n, v, key.repeat(n), mac)
    s0, s1 = additive_split(eng, n, v, gen)
    m0, m1 = additive_split(eng, n, mac, gen)
    if layout == "aos":   # [n][share(4) | mac(4)]
        p0 = torch.cat([s0.view(n, 4), m0.view(n, 4)], dim=1).contiguous().view(-1)
        p1 = torch.cat([s1.view(n, 4), m1.view(n, 4)], dim=1).contiguous().view(-1)
    else:                 # [share column (4n) | mac column (4n)]
        p0 = torch.cat([s0, m0]).contiguous()
        p1 = torch.cat([s1, m1]).contiguous()
    return p0, p1


class Party:
    pass


def build_workload(eng, n, seed, layout, key_shares=None):
    """key_shares: the two parties' MAC key shares (numpy 4 x u64 each) when this batch is one RANGE of a larger one that shares the key
    (the single-process group); by default they are drawn from the seed."""
    gen = torch.Generator(device="cuda")
    gen.manual_seed(seed)
    key_sh = [rand_field_elems(eng, 1, gen), rand_field_elems(eng, 1, gen)]
    if key_shares is not None:
        key_sh = [torch.from_numpy(np.ascontiguousarray(k).view(np.int64)).to(key_sh[0].device) for k in key_shares]
    key = torch.empty_like(key_sh[0])
    eng.scalar_add(1, key_sh[0], key_sh[1], key)
    x = rand_field_elems(eng, n, gen)
    y = rand_field_elems(eng, n, gen)
    a = rand_field_elems(eng, n, gen)
    b = rand_field_elems(eng, n, gen)
    c = torch.empty_like(a)
    eng.scalar_mul(n, a, b, c)
    parties = [Party(), Party()]
    for name, v in (("x", x), ("y", y), ("a", a), ("b", b), ("c", c)):
        p0, p1 = make_shares(eng, n, v, key, gen, layout)
        setattr(parties[0], name, p0)
        setattr(parties[1], name, p1)
    for pid, p in enumerate(parties):
        p.id = pid
        p.key = key_sh[pid].cpu().numpy().view(np.uint64).copy()
        p.de = torch.empty(2 * n * 4, dtype=torch.int64, device="cuda")
        p.out = torch.empty(n * 8, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    return parties, (x, y, key)


def prepare_step(eng, n, parties, layout, chunks=1, k3_order="01"):
    """Pre-bind the launches of a step (arguments marshalled once; buffers are fixed for the whole run).
    With chunks > 1 the batch is cut into gate ranges and each range runs K1(P0), K1(P1), K3(P0), K3(P1)."""
    S = lambda v: ("size", v)
    P = lambda t: t.data_ptr()
    calls = []
    m = n // chunks
    assert m * chunks == n
    for c in range(chunks):
        lo = c * m
        o8, o4 = lo * 64, lo * 32          # byte offsets of gate `lo` in AoS records / 32-byte columns
        de_off = c * 2 * m * 32             # each chunk owns a contiguous d||e block of 2m scalars
        for p in parties:
            if layout == "aos":
                calls.append(eng.prepare("beaver_mask", S(m), P(p.x) + o8, P(p.y) + o8, P(p.a) + o8, P(p.b) + o8, P(p.de) + de_off))
            else:
                calls.append(eng.prepare("beaver_mask_v", S(m), P(p.x) + o4, S(4), P(p.y) + o4, S(4), P(p.a) + o4, S(4), P(p.b) + o4, S(4),
                                         P(p.de) + de_off))
        pairs = ((parties[0], parties[1]), (parties[1], parties[0]))
        for p, peer in (pairs if k3_order == "01" else pairs[::-1]):   # the "network" = reading the peer's d||e
            if layout == "aos":
                calls.append(eng.prepare("beaver_finish_fused", S(m), ("int", p.id), ("key", p.key), P(p.de) + de_off, P(peer.de) + de_off,
                                         P(p.a) + o8, P(p.b) + o8, P(p.c) + o8, P(p.out) + o8))
            else:
                col = 4 * n * 8  # byte offset of the mac column
                calls.append(eng.prepare("beaver_finish_fused_v", S(m), ("int", p.id), ("key", p.key), P(p.de) + de_off, P(peer.de) + de_off,
                                         P(p.a) + o4, P(p.a) + col + o4, S(4), P(p.b) + o4, P(p.b) + col + o4, S(4),
                                         P(p.c) + o4, P(p.c) + col + o4, S(4), P(p.out) + o4, P(p.out) + col + o4, S(4)))
    return calls


def step(calls, eng=None, slot_base=None):
    """One step = the pre-bound launches in order.  With slot_base set, each launch gets a kernel-timer slot: HIP events bound
    to the kernel's own dispatch (hipExtLaunchKernelGGL), so its duration excludes the dispatch gap."""
    if slot_base is None:
        for c in calls:
            c()
        return
    for j, c in enumerate(calls):
        eng.kernel_timer_arm(slot_base + j)
        c()


def check_results(eng, n, parties, truth, layout):
    """open(batch_mul(x, y)) == x*y and the MAC relation holds (reference test_batch_mul, :1571-1594),
    using only engine ops; the bit-exact comparison with the oracle is tests/ and smoke()."""
    x, y, key = truth
    p0, p1 = parties
    if layout == "aos":
        s0, m0 = p0.out.view(n, 8)[:, :4].contiguous().view(-1), p0.out.view(n, 8)[:, 4:].contiguous().view(-1)
        s1, m1 = p1.out.view(n, 8)[:, :4].contiguous().view(-1), p1.out.view(n, 8)[:, 4:].contiguous().view(-1)
    else:
        s0, m0, s1, m1 = p0.out[:4 * n], p0.out[4 * n:], p1.out[:4 * n], p1.out[4 * n:]
    prod = torch.empty_like(x); eng.scalar_mul(n, x, y, prod)
    opened = torch.empty_like(x); eng.scalar_add(n, s0, s1, opened)
    mac = torch.empty_like(x); eng.scalar_add(n, m0, m1, mac)
    kprod = torch.empty_like(x); eng.scalar_mul(n, prod, key.repeat(n), kprod)
    torch.cuda.synchronize()
    return bool(torch.equal(opened, prod)) and bool(torch.equal(mac, kprod))


def host_description():
    """nproc, CPU model, compiler and flags beside the CPU figure; step 1: the cargo probe"""
    import shutil, subprocess
    model = None
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.lower().startswith("model name"):
                model = ln.split(":", 1)[1].strip(); break
    except OSError:
        pass
    def first_line(cmd):
        try:
            return subprocess.run(cmd, capture_output=True, text=True, timeout=20).stdout.strip().splitlines()[0]
        except Exception:            # noqa: BLE001
            return None
    flags = None
    try:
        for ln in open(os.path.join(ROOT, "oracle", "Makefile")):
            if ln.startswith("CFLAGS"):
                flags = ln.split("=", 1)[1].strip(); break
    except OSError:
        pass
    cargo = shutil.which("cargo")
    return {"cpu_model": model, "nproc": os.cpu_count(), "compiler": first_line([os.environ.get("CC", "gcc"), "--version"]), "flags": flags,
            "cargo_probe": (first_line(["cargo", "--version"]) or "present but not runnable") if cargo else "absent (`cargo` not on PATH): the reference's own "
                           "`cargo bench --bench batch_ops` cannot run on this box; the CPU restatement below is timed instead (BASELINE.md section 3 steps 1-2)"}


def cpu_baseline(parties, n, log2n_cpu, layout):
    """the oracle's restatement of the reference's batch_mul (both parties) timed on this host's cores on the first
    2^log2n_cpu gates of the same workload (kind = "port"), in BOTH forms the plan names -- the literal nine passes (authenticated_scalar.rs:
    848-879) and the fused single pass (:799-843 per element), so that the comparison is not hobbled by pass count -- each on all cores (static
    range split in C: the upper bound for the reference's rayon executor) and on one thread (its default single executor thread)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_api
    ora = oracle_api.load()
    m = min(n, 1 << log2n_cpu)

    def host_aos(t):
        if layout == "aos":
            return t[:8 * m].cpu().numpy().view(np.uint64).copy()
        s = t[:4 * m].cpu().numpy().view(np.uint64).reshape(m, 4)
        mm = t[4 * n:4 * n + 4 * m].cpu().numpy().view(np.uint64).reshape(m, 4)
        return np.ascontiguousarray(np.concatenate([s, mm], axis=1).reshape(-1))

    H = [{k: host_aos(getattr(p, k)) for k in "xyabc"} for p in parties]
    keys = [p.key for p in parties]
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    de = [ora.beaver_mask(FID, H[p]["x"], H[p]["y"], H[p]["a"], H[p]["b"]) for p in (0, 1)]   # the peers' d||e: untimed input
    res = [np.zeros(8 * m, dtype=np.uint64) for _ in (0, 1)]
    myde = [np.zeros(8 * m, dtype=np.uint64) for _ in (0, 1)]
    res_f = [np.zeros(8 * m, dtype=np.uint64) for _ in (0, 1)]
    myde_f = [np.zeros(8 * m, dtype=np.uint64) for _ in (0, 1)]
    P = ora._p

    def timed(fn, nthreads, o_de, o_res):
        t0 = time.perf_counter()
        for party in (0, 1):
            h = H[party]
            rc = fn(ctypes.c_int(FID), ctypes.c_size_t(m), ctypes.c_int(party), P(keys[party]), P(h["x"]), P(h["y"]), P(h["a"]), P(h["b"]),
                    P(h["c"]), P(de[1 - party]), P(o_de[party]), P(o_res[party]), ctypes.c_int(nthreads))
            assert rc == 0
        return time.perf_counter() - t0

    def mean_time(fn, nthreads, o_de, o_res, budget_s, max_reps):
        timed(fn, nthreads, o_de, o_res)  # warm
        reps, tot = 0, 0.0
        while tot < budget_s and reps < max_reps:
            tot += timed(fn, nthreads, o_de, o_res); reps += 1
        return tot / reps, reps

    nine, fused = ora.lib.ora_batch_mul_9pass_mt, ora.lib.ora_batch_mul_fused_mt
    t_all, reps = mean_time(nine, cores, myde, res, 4.0, 50)
    t_one, reps1 = mean_time(nine, 1, myde, res, 3.0, 8)
    tf_all, repsf = mean_time(fused, cores, myde_f, res_f, 3.0, 50)
    tf_one, repsf1 = mean_time(fused, 1, myde_f, res_f, 3.0, 8)
    assert np.array_equal(myde[0], de[0]) and np.array_equal(myde[1], de[1])
    same = all(np.array_equal(myde_f[p], myde[p]) and np.array_equal(res_f[p], res[p]) for p in (0, 1))
    assert same, "the fused CPU form disagrees with the nine passes"
    return dict({
        "value": m / t_all, "unit": "gates/s", "cores": cores, "kind": "port",
        "label": "CPU restatement of reference algorithm (not ark-mpc measured)",
        "sample": "first 2^%d gates of the same seeded workload, both parties, the reference's literal 9-pass batch_mul "
                  "(oracle/ark_oracle.c ora_batch_mul_9pass_mt), %d pthreads static range split, mean of %d runs; "
                  "single_thread_value = 1 thread, mean of %d runs; fused_single_pass = the single-gate Mul's closure per element in one sweep "
                  "(ora_batch_mul_fused_mt, authenticated_scalar.rs:799-843), %d / %d runs" % (int(np.log2(m)), cores, reps, reps1, repsf, repsf1),
        "single_thread_value": m / t_one,
        "fused_single_pass": {"value": m / tf_all, "single_thread_value": m / tf_one, "unit": "gates/s", "cores": cores,
                              "same_words_as_nine_passes": bool(same)},
        "excludes": "the reference's DAG-executor overhead (13n+2 result slots per batch_mul, per-argument ResultValue clones, single_threaded.rs:322-356): "
                    "an optimistic stand-in for the reference, i.e. a conservative speed-up denominator (BASELINE.md section 3 step 3)",
    }, **host_description()), res, myde, m


def run_pipeline(eng, n, sets, layout, args, steps, warmup, barrier, settle_ms=None, rounds=1):
    """`warmup` untimed and `steps` timed passes of the pipeline over the rotated workload sets.  The timed region is
    bracketed by barrier() (dist.barrier + torch.cuda.synchronize) on both sides; on sampled steps every launch carries a
    dispatch-bound HIP event pair (arkmpc_kernel_timer_*, on the context's own stream = torch's current stream)."""
    chunks = args.chunks if args.chunks > 0 else max(1, n >> 20)
    call_sets = [prepare_step(eng, n, ps, layout, chunks, args.k3_order) for ps, _ in sets]
    per_step = 4 * chunks                           # launches per step: per gate range K1(P0), K1(P1), K3(P0), K3(P1)
    barrier()                                       # the FIRST barrier of a process group builds the RCCL communicator (100s of ms with an idle GPU): pay that
                                                    # here, before the settle / warm-up phases, so the barrier that opens the timed region is only a barrier
    settle_ms = getattr(args, "settle_ms", 0) if settle_ms is None else settle_ms
    if settle_ms > 0:                               # disclosed in config.settle_ms: steady-state clocks before the warm-up steps
        t_s = time.perf_counter()
        k = 0
        while (time.perf_counter() - t_s) * 1e3 < settle_ms:
            for _ in range(8):
                step(call_sets[k % len(call_sets)]); k += 1
            torch.cuda.synchronize()
    for w in range(warmup):
        step(call_sets[w % len(call_sets)])
    barrier()
    max_sampled = min(16, 64 // per_step)            # the engine has 64 kernel-timer slots
    every = max(1, args.event_every, -(-steps // max(1, max_sampled)))
    sampled = [s for s in range(steps) if s % every == 0][:max_sampled]
    slot_of = {s: per_step * i for i, s in enumerate(sampled)}
    ev_begin, ev_end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev_begin.record()
    for s in range(steps):                          # round 0: sampled steps carry per-kernel events
        step(call_sets[s % len(call_sets)], eng, slot_of.get(s))
    for r in range(1, rounds):                      # the same K steps again, until the region is long enough to time (--min-timed-ms)
        for s in range(steps):
            step(call_sets[s % len(call_sets)])
    ev_end.record()
    barrier()
    elapsed = time.perf_counter() - t0
    if sampled:
        seg = np.array([[eng.kernel_timer_ms(slot_of[s] + j) for j in range(per_step)] for s in sampled]).reshape(len(sampled), chunks, 4)  # ms
        k1_ms, k3_ms = float(seg[:, :, :2].mean()), float(seg[:, :, 2:].mean())
    else:
        k1_ms = k3_ms = float("nan")
    return {"elapsed": elapsed, "k1_ms": k1_ms, "k3_ms": k3_ms, "dev_ms_per_step": ev_begin.elapsed_time(ev_end) / (steps * rounds),
            "chunks": chunks, "sampled": len(sampled), "rounds": rounds}


def oracle_bitexact(parties, n, m, chunks, layout, res, myde):
    """Word-for-word comparison of the GPU buffers of the timed workload (both parties: own d||e and the result records)
    with what the oracle computed for the first m gates.  Returns the number of gates on which EVERY word matched."""
    mc = n // chunks
    ok = np.ones(m, dtype=bool)
    for pid, p in enumerate(parties):
        out = p.out.cpu().numpy().view(np.uint64)
        if layout == "aos":
            got = out[:8 * m].reshape(m, 8)
        else:
            got = np.concatenate([out[:4 * m].reshape(m, 4), out[4 * n:4 * n + 4 * m].reshape(m, 4)], axis=1)
        ok &= (got == res[pid].reshape(m, 8)).all(axis=1)
        de = p.de.cpu().numpy().view(np.uint64).reshape(chunks, 2, mc, 4)          # per gate range: d block, then e block
        d, e = de[:, 0].reshape(n, 4)[:m], de[:, 1].reshape(n, 4)[:m]
        ok &= (d == myde[pid][:4 * m].reshape(m, 4)).all(axis=1) & (e == myde[pid][4 * m:].reshape(m, 4)).all(axis=1)
    return int(ok.sum())


def timed_events(fn, reps, warm=1):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps           # ms


def leg_aos(eng, n, args):
    """The arkworks record layout exactly as a Rust caller holds it (Vec<ScalarShare>, 64-byte records), two ways:
    (i) the AoS entry points directly; (ii) a cold AoS caller of the engine-native path: arkmpc_share_split of x, y, a, b, c,
    the split-column pipeline, arkmpc_share_join of the result -- every step, nothing kept resident."""
    sets = [build_workload(eng, n, seed=0xA11CE0A0 + 7919 * k, layout="aos") for k in range(2)]
    steps = max(1, min(args.steps, 100))
    r = run_pipeline(eng, n, sets, "aos", args, steps, min(args.warmup, 10), torch.cuda.synchronize)
    ok = all(check_results(eng, n, ps, tr, "aos") for ps, tr in sets[:min(len(sets), steps)])
    out = {"layout": "arkworks AoS ScalarShare records (64 B), consumed as they lie", "gates_per_s": n * steps / r["elapsed"],
           "device_ms_per_step": r["dev_ms_per_step"], "k1_avg_launch_ms": r["k1_ms"], "k3_avg_launch_ms": r["k3_ms"],
           "pipeline_frac_of_hbm_peak": n * ALG_BYTES_PER_GATE / (r["dev_ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBPS,
           "actual_bytes_per_party_gate": 704, "results_check": "ok" if ok else "FAILED",
           "note": "floor: K1 must fetch whole 64 B x,y,a,b records for their share halves (320 B read + 64 written) and K3 re-reads a,b "
                   "(384 B) = 704 B per party-gate vs 512 B algorithmic"}
    tf = os.path.join(ROOT, "profiles", "traffic_aos.json")
    if os.path.exists(tf):
        out["traffic_source"] = "profiles/traffic_aos.json (committed rocprofv3 PMC passes, not measured in this run)"
        out["traffic"] = json.load(open(tf))
    # (ii) cold AoS caller through the split path
    parties = sets[0][0]
    S, P = (lambda v: ("size", v)), (lambda t: t.data_ptr())
    calls = []
    cols = []
    for p in parties:
        c = {k: torch.empty(8 * n, dtype=torch.int64, device="cuda") for k in "xyabco"}
        cols.append(c)
    col = 4 * n * 8
    for p, c in zip(parties, cols):
        for k in "xyab":
            calls.append(eng.prepare("share_split", S(n), P(getattr(p, k)), P(c[k]), P(c[k]) + col))
        calls.append(eng.prepare("beaver_mask_v", S(n), P(c["x"]), S(4), P(c["y"]), S(4), P(c["a"]), S(4), P(c["b"]), S(4), P(p.de)))
    for (p, c), peer in zip(zip(parties, cols), parties[::-1]):
        calls.append(eng.prepare("share_split", S(n), P(p.c), P(c["c"]), P(c["c"]) + col))
        calls.append(eng.prepare("beaver_finish_fused_v", S(n), ("int", p.id), ("key", p.key), P(p.de), P(peer.de),
                                 P(c["a"]), P(c["a"]) + col, S(4), P(c["b"]), P(c["b"]) + col, S(4), P(c["c"]), P(c["c"]) + col, S(4),
                                 P(c["o"]), P(c["o"]) + col, S(4)))
        calls.append(eng.prepare("share_join", S(n), P(c["o"]), P(c["o"]) + col, P(p.out)))
    ms = timed_events(lambda: [c() for c in calls], reps=20, warm=3)
    ok2 = check_results(eng, n, parties, sets[0][1], "aos")
    out["cold_caller_via_split_import"] = {"gates_per_s": n / (ms * 1e-3), "device_ms_per_step": ms, "results_check": "ok" if ok2 else "FAILED",
                                           "what": "per step and party: share_split(x,y,a,b,c) + K1 + K2+K3 on columns + share_join(result)"}
    return out, ok and ok2


# Integer-ALU accounting of a BN254 G1 scalar-mul.  The shipped path is the hand-scheduled pipeline (tools/gen_ec_asm.py); the generator
# counts the multiplier instructions (v_mad_u64_u32 + v_mul_lo_u32) one scalar-mul executes in its two asm kernels and writes them to
# csrc/ec_asm_stats.json.  frac_of_int_alu_peak = those instructions per second / the measured chip-wide v_mad_u64_u32 rate: a true
# utilisation.  The round-1 figure counted 2004 general multiplications of 136 multiplier instructions for the then algorithm (GLV, signed
# 5-bit windows, Jacobian table); it is kept as `r01_accounting` so the two rounds can be compared on equal work.
MAD_PEAK_PER_S = 31.2e12        # v_mad_u64_u32 lane-ops/s chip-wide, measured (profiles/ubench_r01.log)
VALU_NOMINAL_PER_S = 256 * 4 * 16 * 2.4e9     # nominal VALU issue rate: 256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz = 39.3e12 lane-ops/s
MADS_PER_FQ_MUL = 136           # 64 product + 64 reduction v_mad_u64_u32 + 8 v_mul_lo_u32 (the m = t0 * inv words)
FQ_MULS_PER_SMUL_R01 = 27 * (5 * 7 + 2 * 16 + 1) + 8 * 7 + 7 * 16


def ec_limbs():
    return 32 if os.environ.get("ARKMPC_EC_LIMBS") == "32" else 29


def ec_mult_instrs():
    """the shipped kernels compute on nine 29-bit limbs (tools/gen_ec29_asm.py); ARKMPC_EC_LIMBS=32 selects the round-2 32-bit-limb ones"""
    st = json.load(open(os.path.join(ROOT, "ark-mpc_amd", "csrc", "ec29_asm_stats.json" if ec_limbs() == 29 else "ec_asm_stats.json")))
    return st["mult_instrs_loop"] + st["mult_instrs_table"], st


def leg_config4(eng):
    """BASELINE config 4: 2^18 PointShare x public Scalar over BN254 G1 (curve/share.rs:108-114) = 2^19 scalar-muls.
    Points are k_i * G with known k_i, so the result is checked against the fixed-base path [(s_i k_i)]G on affine coordinates."""
    n = 1 << 18
    gen = torch.Generator(device="cuda"); gen.manual_seed(0xA11CE004)
    k = rand_field_elems(eng, 2 * n, gen)                   # n ScalarShares: the discrete logs of (share, mac)
    shares = torch.empty(24 * n, dtype=torch.int64, device="cuda")
    eng.scalarshare_mul_generator(n, k, shares)
    sc = rand_field_elems(eng, n, gen)
    out = torch.empty_like(shares)
    ms = timed_events(lambda: eng.pointshare_mul_public(n, shares, sc, out), reps=5, warm=1)
    sk = torch.empty_like(k)
    eng.scalar_mul(2 * n, k, sc.view(n, 1, 4).expand(n, 2, 4).contiguous().view(-1), sk)
    want = torch.empty(12 * 2 * n, dtype=torch.int64, device="cuda")
    eng.g1_generator_mul(2 * n, sk, want)
    xy = [torch.empty(8 * 2 * n, dtype=torch.int64, device="cuda") for _ in (0, 1)]
    inf = [torch.empty(2 * n, dtype=torch.uint8, device="cuda") for _ in (0, 1)]
    eng.g1_to_affine(2 * n, out, xy[0], inf[0]); eng.g1_to_affine(2 * n, want, xy[1], inf[1])
    torch.cuda.synchronize()
    ok = bool(torch.equal(xy[0], xy[1])) and bool(torch.equal(inf[0], inf[1]))
    # the fixed-base chain shares the hand-scheduled mixed-addition body with the variable-base pipeline, so it is not an independent
    # witness: a sample of lanes is also compared with the oracle's double-and-add (oracle/ark_oracle.c, a checker outside the timed region)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_api
    ora = oracle_api.load()
    ns = 1024
    h_sh = shares[:24 * ns].cpu().numpy().view(np.uint64).copy()
    h_sc = sc[:4 * ns].cpu().numpy().view(np.uint64).copy()
    want_o = ora.pointshare_mul_public_mt(h_sh, h_sc)
    oxy, oinf = ora.g1_batch_to_affine_mt(want_o)
    ok_oracle = bool(np.array_equal(oxy, xy[0][:16 * ns].cpu().numpy().view(np.uint64))) and bool(np.array_equal(oinf, inf[0][:2 * ns].cpu().numpy()))
    ok = ok and ok_oracle
    smuls = 2 * n / (ms * 1e-3)
    per_smul, st = ec_mult_instrs()
    # the same call on HOST vectors (what a gate closure holds): PointShares 48 MiB + scalars 8 MiB up, PointShares 48 MiB down.  Never `ms` above.
    host = {"what": "arkmpc_pointshare_mul_public on a host-buffer context (arkmpc_ctx_set_host_buffers), numpy vectors in and out: staged whole "
                    "(upload, the four kernels, download); 104 MiB over the link = 1.9 ms of the figure"}
    try:
        pkg_ = importlib.import_module("ark-mpc_amd")
        lib_ = pkg_.load_library()
        eh = pkg_.Engine(FID, device=torch.cuda.current_device(), host_buffers=True)
        h_in, h_s = shares.cpu().numpy().view(np.uint64).copy(), sc.cpu().numpy().view(np.uint64).copy()
        h_out = np.zeros_like(h_in)
        want_h = out.cpu().numpy().view(np.uint64)

        def timed_host(reps=4):
            eh.pointshare_mul_public(n, h_in, h_s, h_out)
            t0 = time.perf_counter()
            for _ in range(reps):
                eh.pointshare_mul_public(n, h_in, h_s, h_out)
            return (time.perf_counter() - t0) / reps * 1e3
        host["pageable_ms"] = timed_host()
        ok_h = bool(np.array_equal(h_out, want_h))
        for a_ in (h_in, h_s, h_out):
            lib_.arkmpc_host_register(ctypes.c_void_p(a_.ctypes.data), ctypes.c_size_t(a_.nbytes))
        h_out.fill(0)
        host["registered_ms"] = timed_host()
        ok_h = ok_h and bool(np.array_equal(h_out, want_h))
        for a_ in (h_in, h_s, h_out):
            lib_.arkmpc_host_unregister(ctypes.c_void_p(a_.ctypes.data))
        eh.close()
        host["check"] = "every word == the device-resident call's result: %s" % ("ok" if ok_h else "FAILED")
        ok = ok and ok_h
    except Exception as ex:      # noqa: BLE001
        host["error"] = repr(ex)[:200]
        ok = False
    return {"workload": "2^18 PointShare x Scalar over BN254 G1 = 2^19 scalar-muls (BASELINE.json configs[3])", "ms": ms, "host_vectors": host,
            "secondary_op": config4_secondary(),
            "scalar_muls_per_s": smuls, "bound": "integer ALU",
            "algorithm": "GLV + signed 5-bit windows; effective-affine window table (common Z), blinded accumulator, mixed additions; digits / table / "
                         "window loop / finish kernels, table + loop hand-scheduled on %s" % (
                             "nine unsaturated 29-bit limbs, product-scanning Montgomery multiplier with one 64-bit column accumulator (tools/gen_ec29_asm.py)"
                             if ec_limbs() == 29 else "eight 32-bit limbs, CIOS rows (tools/gen_ec_asm.py)"),
            "limbs": ec_limbs(),
            "mult_instrs_per_scalar_mul": per_smul, "mult_instrs_per_s": smuls * per_smul,
            "frac_of_int_alu_peak": smuls * per_smul / MAD_PEAK_PER_S,
            "frac_of_nominal_valu_rate": smuls * per_smul / VALU_NOMINAL_PER_S,
            "int_alu_peak_note": "multiplier instructions (v_mad_u64_u32 + v_mul_lo_u32) executed per second, against two denominators: frac_of_int_alu_peak = / 31.2e12 "
                                 "lane-ops/s, the chip-wide v_mad_u64_u32 rate MEASURED on this part (probes/ubench.hip); frac_of_nominal_valu_rate = / 39.3e12, the nominal "
                                 "VALU issue rate (256 CU x 4 SIMD x 16 lanes x 2.4 GHz), which no multiplier stream reaches",
            "ceiling_note": ("PMC (profiles/r03_ec/pmc_limbs29.txt): 4.08 SIMD cycles per VALU instruction in loop and table -- the issue limit; "
                             "70 % of the instructions are multiplier instructions") if ec_limbs() == 29 else
                            ("a bare chain of the hand-scheduled Montgomery block reaches 0.61 of this peak (probes/mulrate.hip, profiles/r02/mulrate.jsonl): "
                             "162 of its 298 instructions are carries and moves"),
            "r01_accounting": {"fq_muls_per_scalar_mul": FQ_MULS_PER_SMUL_R01, "fq_muls_per_s": smuls * FQ_MULS_PER_SMUL_R01,
                               "frac_of_mad_only_peak": smuls * FQ_MULS_PER_SMUL_R01 * MADS_PER_FQ_MUL / MAD_PEAK_PER_S,
                               "note": "round 1's work definition (2004 general multiplications per scalar-mul) at this round's speed"},
            "results_check": "affine coords == fixed-base [(s*k)]G on all 2^19 points, and == the oracle's double-and-add on the first 2048 scalar-muls: %s" % ("ok" if ok else "FAILED")}, ok


def config4_secondary():
    """BASELINE config 4's secondary op, AuthenticatedPointResult::batch_mul (authenticated_curve.rs:682-714) at 2^18, through the C++ host
    mirror (two parties in one process, dummy Beaver source, device link): the mirror's own bench binary, run as a subprocess."""
    import subprocess
    exe = os.path.join(ROOT, "ark-mpc_amd", "lib", "arkmpc_host_bench")
    if not os.path.exists(exe):
        return {"note": "arkmpc_host_bench not built"}
    out = {}
    for name, literal in (("regrouped", "0"), ("literal_sequence", "1")):
        try:
            r = subprocess.run([exe, "point_batch_mul", str(1 << 18), "2"], capture_output=True, text=True, timeout=120,
                               env=dict(os.environ, ARKMPC_MOCK_LINK="device", ARKMPC_POINT_MUL_LITERAL=literal))
            d = json.loads(r.stdout.strip().splitlines()[-1])
            out[name] = {"ms_both_parties": d["seconds"] * 1e3, "elements_per_s": d["elements_per_s"]}
        except Exception as ex:      # noqa: BLE001
            out[name] = {"error": repr(ex)[:200]}
    out["what"] = ("[x * yG] by a Beaver triple for 2^18 elements, both parties on one GPU; regrouped = ([a]+d) eG + ([c]+d[b]) G, 2 variable-base + 4 generator "
                   "scalar-muls per element and party (the form the engine's host mirror runs); literal_sequence = the reference's 6 + 4; "
                   "equal share by share (tests/test_host_fabric.py::test_point_beaver_mul_regrouped_equals_literal_sequence)")
    return out


def leg_config5(pkg, dev):
    """BASELINE config 5 shape on ONE GPU: open_authenticated_batch (authenticated_scalar.rs:278-354) over 2^24 BLS12-381 Fr
    shares, both parties in-process.  Device part: `.share()` extraction, K2+K4, K5 for both parties; host part: each party
    hashes two 512 MiB streams (its own commitment, the peer's for verification) -- sequential sponges by the reference's
    definition; the two parties hash concurrently, a party's own two sponges are ordered by the protocol."""
    import threading
    fid, n = 1, 1 << 24
    eng = pkg.Engine(fid, device=dev, stream=torch.cuda.current_stream().cuda_stream)
    gen = torch.Generator(device="cuda"); gen.manual_seed(0xA11CE005)
    ks = [rand_field_elems(eng, 1, gen), rand_field_elems(eng, 1, gen)]
    key = torch.empty_like(ks[0]); eng.scalar_add(1, ks[0], ks[1], key)
    keys = [t.cpu().numpy().view(np.uint64).copy() for t in ks]
    v = rand_field_elems(eng, n, gen)
    sh = list(make_shares(eng, n, v, key, gen, "aos"))
    mine = [torch.empty(4 * n, dtype=torch.int64, device="cuda") for _ in (0, 1)]
    opened = [torch.empty(4 * n, dtype=torch.int64, device="cuda") for _ in (0, 1)]
    chk = [torch.empty(4 * n, dtype=torch.int64, device="cuda") for _ in (0, 1)]
    blind = [rand_field_elems(eng, 1, gen).cpu().numpy().view(np.uint64).copy() for _ in (0, 1)]
    oks = []

    def device_part():
        for p in (0, 1):
            eng.share_extract(n, sh[p], mine[p])
        for p in (0, 1):
            eng.open_and_mac_check(n, keys[p], sh[p], mine[1 - p], opened[p], chk[p])
        oks[:] = [eng.mac_verify(n, chk[p], chk[1 - p]) for p in (0, 1)]

    ms_dev = timed_events(device_part, reps=5, warm=1)
    ok = oks == [True, True] and bool(torch.equal(opened[0], v)) and bool(torch.equal(opened[1], v))
    # the same step on the engine-native split columns (shares resident as gate outputs are kept): the payload a party sends IS its share
    # column -- no extraction pass -- and the MAC half is read once: 160 + 64 = 224 B per party-share of traffic instead of 96 + 160 + 64 = 320
    cols = []
    for p in (0, 1):
        sc, mc = torch.empty(4 * n, dtype=torch.int64, device="cuda"), torch.empty(4 * n, dtype=torch.int64, device="cuda")
        eng.share_split(n, sh[p], sc, mc)
        cols.append((sc, mc))
    oks2 = []

    def device_part_split():
        for p in (0, 1):
            eng.open_and_mac_check_v(n, keys[p], cols[p][0], cols[p][1], 4, cols[1 - p][0], opened[p], chk[p])
        oks2[:] = [eng.mac_verify(n, chk[p], chk[1 - p]) for p in (0, 1)]

    opened[0].zero_(); opened[1].zero_()
    ms_dev_split = timed_events(device_part_split, reps=5, warm=1)
    ok = ok and oks2 == [True, True] and bool(torch.equal(opened[0], v)) and bool(torch.equal(opened[1], v))
    del cols
    t0 = time.perf_counter()
    c_one = eng.commit_sha3(n, chk[0], blind[0])
    ms_one = (time.perf_counter() - t0) * 1e3
    # end to end: device part, then the sponges in the order the protocol allows.  A party's two sponges cannot overlap: its own
    # commitment must be sent BEFORE the peer reveals its MAC-check shares (commit-then-reveal, authenticated_scalar.rs:313-340),
    # and the second sponge hashes exactly those revealed shares (commitment.rs:30-43).  The two PARTIES do run concurrently
    # (one host thread and one context each): phase 1 = both commit, phase 2 = both re-hash the peer's shares.
    ctxs = [pkg.Engine(fid, device=dev) for _ in range(2)]
    comm = [None] * 4

    def sponge(slot, party, which):
        torch.cuda.set_device(dev)
        comm[slot] = ctxs[party].commit_sha3(n, chk[which], blind[which])

    torch.cuda.synchronize()
    t0 = time.perf_counter()
    device_part()
    torch.cuda.synchronize()
    for phase in (0, 1):                     # phase 0: commit(own chk); phase 1: verify = hash(peer chk, peer blinder)
        th = [threading.Thread(target=sponge, args=(2 * phase + p, p, p if phase == 0 else 1 - p)) for p in (0, 1)]
        for t in th: t.start()
        for t in th: t.join()
    ms_e2e = (time.perf_counter() - t0) * 1e3
    ok = ok and np.array_equal(comm[0], c_one) and np.array_equal(comm[0], comm[3]) and np.array_equal(comm[1], comm[2])
    for c in ctxs: c.close()
    eng.close()
    return {"workload": "open_authenticated_batch over 2^24 BLS12-381 Fr shares, both parties on one GPU (BASELINE.json configs[4] shape)",
            "device_ms_both_parties": ms_dev, "device_what": "share extract + K2+K4 (open + MAC-check shares) + K5 (verify) for both parties",
            "device_shares_per_s": n / (ms_dev * 1e-3), "device_alg_GBps": 2 * n * 256 / (ms_dev * 1e-3) / 1e9,
            "device_frac_of_hbm_peak": 2 * n * 256 / (ms_dev * 1e-3) / 1e9 / HBM_PEAK_GBPS, "alg_bytes_per_party_share": 256,
            "layout": "arkworks AoS ScalarShare records (what the boundary receives)",
            "split_columns": {"device_ms_both_parties": ms_dev_split, "device_shares_per_s": n / (ms_dev_split * 1e-3),
                              "device_alg_GBps": 2 * n * 256 / (ms_dev_split * 1e-3) / 1e9,
                              "device_frac_of_hbm_peak": 2 * n * 256 / (ms_dev_split * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                              "note": "shares resident in the engine-native split columns: no extraction pass (the share column is the payload), K2+K4 column form + K5; "
                                      "224 B of traffic per party-share against the 256 B algorithmic figure, which counts the payload write"},
            "host_sha3_ms_one_commitment": ms_one, "host_sha3_MBps": 32 * n / (ms_one * 1e-3) / 1e6,
            "host_sha3_note": "one sequential SHA3-256 over 512 MiB (commitment.rs:36-40 hashes one message); 4 such per batch (2 per party)",
            "end_to_end_ms": ms_e2e, "host_sha3_share_of_end_to_end": max(0.0, 1.0 - ms_dev / ms_e2e),
            "lead": "end to end this configuration is host SHA3: %.0f ms of %.0f ms (%.1f %%) are the four sequential sponges (commitment.rs:36-40 hashes ONE message per "
                    "commitment); the device part is %.2f ms, so the device fractions below describe a stage nobody waits for" % (ms_e2e - ms_dev, ms_e2e, 100 * (1 - ms_dev / ms_e2e), ms_dev),
            "end_to_end_what": "device part + commit phase + verify phase; the two parties hash concurrently (one host thread each), "
            "a party's own two sponges are ordered by the commit-then-reveal protocol and cannot overlap",
            "results_check": "opened == value on all shares, both MAC checks verify, each recomputed commitment == the peer's: %s" % ("ok" if ok else "FAILED")}, ok


E2E_UP_BYTES, E2E_DOWN_BYTES = 384, 128     # per party-gate over the host link: x, y, a, b, c records + the peer's d||e up; own d||e + result record down


def pcie_calibration(mib=256):
    """What the host link of this box gives plain pinned copies (the ceiling of the streaming path): one direction, and both at once."""
    m = mib << 20
    dev, dev2 = torch.empty(m, dtype=torch.uint8, device="cuda"), torch.empty(m, dtype=torch.uint8, device="cuda")
    h, h2 = torch.empty(m, dtype=torch.uint8).pin_memory(), torch.empty(m, dtype=torch.uint8).pin_memory()
    h.fill_(1); h2.fill_(2)
    s2 = torch.cuda.Stream()

    def both():
        dev.copy_(h, non_blocking=True)
        with torch.cuda.stream(s2):
            h2.copy_(dev2, non_blocking=True)

    out = {}
    for name, fn, vol in (("h2d", lambda: dev.copy_(h, non_blocking=True), m), ("d2h", lambda: h2.copy_(dev2, non_blocking=True), m), ("both", both, 2 * m)):
        fn(); torch.cuda.synchronize()
        best = 0.0
        for _ in range(6):                               # best of six batches of four copies (the denominator of frac_of_measured_pcie: a low reading would flatter the path)
            t0 = time.perf_counter()
            for _ in range(4):
                fn()
            torch.cuda.synchronize()
            best = max(best, vol / ((time.perf_counter() - t0) / 4) / 1e9)
        out[name + "_GBps"] = best
    return out


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

    pageable = timed_one("pageable, NEW vectors for every session (numpy / Vec memory); pinned in place inside each call and moved by DMA (no kernel addresses a vector the library registered itself, DESIGN section 4)", True)
    regs = [a for p in (0, 1) for a in list(H[p].values())] + de + out + want_de
    for a in regs:
        lib.arkmpc_host_register(ctypes.c_void_p(a.ctypes.data), ctypes.c_size_t(a.nbytes))
    registered = timed_one("registered once by the caller (arkmpc_host_register), as a caller that keeps its vectors across gates would: both phases run as kernels that read and write "
                           "the pinned vectors in place, no copy commands (ARKMPC_HOSTMUL_ZEROCOPY=0 puts them back on the copy pipeline: 8.0-8.1 ms at 2^20)", False)
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
    try:
        cap = eng.wire_frame_bound(2 * n)
        fbuf = []
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
    # the oracle on a sample of the same host data (the device-resident buffers used as the expectation above are not an independent witness)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_api
    ora = oracle_api.load()
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    m = n if cores >= 16 else min(n, 1 << 16)          # every gate where the host has the threads for it (the GPU boxes: 256), a 2^16 sample elsewhere
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
    return {"what": "host arkworks records in -> host records out, 2^%d Beaver muls over BN254 Fr per party (benches/batch_ops.rs shape); NOT the metric's `value`, "
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


def _pinned_array(lib, nwords):
    """a numpy u64 array over an arkmpc_host_alloc block (pinned, recycled); returns (array, pointer)"""
    q = ctypes.c_void_p()
    if lib.arkmpc_host_alloc(ctypes.c_size_t(8 * nwords), ctypes.byref(q)) != 0:
        raise RuntimeError("arkmpc_host_alloc(%d bytes)" % (8 * nwords))
    return np.ctypeslib.as_array(ctypes.cast(q, ctypes.POINTER(ctypes.c_uint64)), shape=(nwords,)), q


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
        arr, q = _pinned_array(lib, 8 * n)
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
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_api
    ora = oracle_api.load()
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    m = n if cores >= 16 else min(n, 1 << 16)
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
    return {"what": "depth-%d chain z <- z * y of batch_mul gates at 2^%d, operands resident (split columns), both parties on this ONE GPU and its one host link, d||e handed "
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


def clock_effect():
    """Measured effect of the profiler on the dominant kernel, from the committed PMC pass (profiles/r0N/clock_effect.json, written by
    tools/profile.sh): GRBM_GUI_ACTIVE cycles / the kernel's wall time under rocprofv3 = the shader clock it ran at while profiled."""
    for rnd in ("r05", "r04", "r03"):
        f = os.path.join(ROOT, "profiles", rnd, "clock_effect.json")
        if os.path.exists(f):
            d = json.load(open(f)); d["source"] = "profiles/%s/clock_effect.json" % rnd
            return d
    return None


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
        a_, q = _pinned_array(lib, arr.size)
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
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_api
    ora = oracle_api.load()
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    m_ = n if cores >= 16 else min(n, 1 << 16)
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


def rank_identity(dev):
    """what identifies the physical GPU this rank computes on: gathered over the process group into the N>1 line, so that `N ranks on N distinct
    devices` can be read off the line itself"""
    pr = torch.cuda.get_device_properties(dev)
    ident = {"local_device": int(dev), "name": pr.name}
    for k in ("uuid", "pci_bus_id", "pci_device_id", "pci_domain_id"):
        if hasattr(pr, k):
            v = getattr(pr, k)
            ident[k] = v if isinstance(v, int) else str(v)
    ident["pid"] = os.getpid()
    return ident


def per_rank_oracle_check(parties, n, chunks, layout, m=1 << 12):
    """every rank checks the first 2^12 gates of ITS timed buffers (both parties: d||e and result records) against the oracle -- N>1 runs are
    not parity-blind.  Returns the number of gates on which every word matched (m = all)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_api
    ora = oracle_api.load()
    m = min(m, n // chunks)

    def host_aos(t):
        if layout == "aos":
            return t[:8 * m].cpu().numpy().view(np.uint64).copy()
        sh = t[:4 * m].cpu().numpy().view(np.uint64).reshape(m, 4)
        mm = t[4 * n:4 * n + 4 * m].cpu().numpy().view(np.uint64).reshape(m, 4)
        return np.ascontiguousarray(np.concatenate([sh, mm], axis=1).reshape(-1))

    H = [{k: host_aos(getattr(p, k)) for k in "xyabc"} for p in parties]
    ode = [ora.beaver_mask(FID, H[p]["x"], H[p]["y"], H[p]["a"], H[p]["b"]) for p in (0, 1)]
    res, myde = [], []
    for p in (0, 1):
        d_, r_ = ora.batch_mul_9pass_local(FID, p, parties[p].key, H[p]["x"], H[p]["y"], H[p]["a"], H[p]["b"], H[p]["c"], ode[1 - p])
        myde.append(d_); res.append(r_)
    return oracle_bitexact(parties, n, m, chunks, layout, res, myde), m


def main():
    args = parse()
    if args.single_process:
        return main_single_process(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU fallback")
    if world > 1 or "TORCHELASTIC_RUN_ID" in os.environ:     # under torch.distributed.run the collective path is used even for one rank
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        local_rank = local_rank % torch.cuda.device_count()      # (tests may oversubscribe one GPU with the gloo backend)
        torch.cuda.set_device(local_rank)
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.dist_backend)
    else:
        dist = None
        torch.cuda.set_device(0)
    dev = torch.cuda.current_device()
    cdev = "cuda" if args.dist_backend == "nccl" else "cpu"       # where collective tensors live
    pkg = importlib.import_module("ark-mpc_amd")
    eng = pkg.Engine(FID, device=dev, host_buffers=False, stream=torch.cuda.current_stream().cuda_stream)
    if args.only_circuit:
        r, ok = leg_circuit(pkg, eng, dev, args.circuit_log2n, args.circuit_depth)
        print(json.dumps(r), flush=True)
        eng.close()
        if not ok:
            raise SystemExit("result check failed")
        return
    if args.only_e2e:
        r, ok = leg_end_to_end(pkg, eng, dev, args.e2e_log2n)
        print(json.dumps(r), flush=True)
        eng.close()
        if not ok:
            raise SystemExit("result check failed")
        return
    if args.scaling == "strong":                   # fixed total work: 2^total_log2n gates per step cut into `world` contiguous ranges
        if (1 << args.total_log2n) % world:
            raise SystemExit("--scaling strong needs a power-of-two number of GPUs")
        n = (1 << args.total_log2n) // world
        args.log2n = int(np.log2(n))
    else:
        if args.log2n is None:
            args.log2n = 21 if world == 8 else 20      # 8 ranks: BASELINE config 3 (2^24 gates over 8 GPUs)
        n = 1 << args.log2n
    sets = [build_workload(eng, n, seed=0xA11CE002 + rank + 7919 * k, layout=args.layout) for k in range(max(1, args.sets))]
    parties, truth = sets[0]

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def reduce_over_ranks(v, op):
        if dist is None:
            return v
        t = torch.tensor([v], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=op)
        return float(t.item())

    def max_over_ranks(v):
        return v if dist is None else reduce_over_ranks(v, dist.ReduceOp.MAX)

    # who is here: every rank adds 1 over the collective backend (RCCL under the driver) and contributes the identity of its GPU
    ranks_seen, idents = 1, [rank_identity(dev)]
    if dist is not None:
        ranks_seen = int(round(reduce_over_ranks(1.0, dist.ReduceOp.SUM)))
        idents = [None] * world
        dist.all_gather_object(idents, rank_identity(dev))

    # cold pass FIRST: exactly W warm-up + K timed steps with no settle phase -- what a K-step region measures on a GPU that was idle while
    # the workload was built (the power controller's transient, profiles/r02/ramp_probe.txt).  Reported beside the headline; its duration
    # also sizes the number of rounds of the headline region.
    cold = None
    if not args.no_cold and args.settle_ms > 0:
        rc_ = run_pipeline(eng, n, sets, args.layout, args, args.steps, args.warmup, barrier, settle_ms=0)
        cold = {"elapsed": max_over_ranks(rc_["elapsed"]), "k3_ms": rc_["k3_ms"], "k1_ms": rc_["k1_ms"], "dev_ms_per_step": rc_["dev_ms_per_step"]}
        est_region_ms = cold["elapsed"] * 1e3
    else:
        rc_ = run_pipeline(eng, n, sets, args.layout, args, min(8, args.steps), 2, barrier, settle_ms=0)
        est_region_ms = max_over_ranks(rc_["elapsed"]) * 1e3 * args.steps / min(8, args.steps)
    rounds = 1
    if args.min_timed_ms > 0 and est_region_ms > 0:
        rounds = max(1, int(np.ceil(1.2 * args.min_timed_ms / est_region_ms)))     # (the sizing pass runs cold and a little slow: 20 % margin)
    rounds = int(max_over_ranks(float(rounds)))       # one number on every rank
    r = run_pipeline(eng, n, sets, args.layout, args, args.steps, args.warmup, barrier, rounds=rounds)
    k1_ms, k3_ms, dev_ms_per_step, chunks = r["k1_ms"], r["k3_ms"], r["dev_ms_per_step"], r["chunks"]
    own_elapsed = r["elapsed"]
    elapsed = max_over_ranks(own_elapsed)
    steps_total = args.steps * rounds
    per_rank_ms = [own_elapsed / steps_total * 1e3]
    if dist is not None:
        t = torch.tensor([own_elapsed / steps_total * 1e3], dtype=torch.float64, device=cdev)
        allt = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(allt, t)
        per_rank_ms = [float(x.item()) for x in allt]

    ok = True if args.no_check else all(check_results(eng, n, ps, tr, args.layout) for ps, tr in sets[:min(len(sets), args.steps)])
    # parity on every rank (N = 1 runs the full-size comparison below instead)
    rank_exact = None
    if dist is not None and not args.no_check:
        # the checker library is (re)built by `make` on first load: one rank of the node does that, the others load the finished file
        if local_rank == 0:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import oracle_api
            oracle_api.load()
        dist.barrier()
        exact_r, m_r = per_rank_oracle_check(parties, n, chunks, args.layout)
        ok = ok and exact_r == m_r
        rank_exact = {"gates_checked_per_rank": m_r, "ranks_all_exact": bool(reduce_over_ranks(1.0 if exact_r == m_r else 0.0, dist.ReduceOp.MIN) == 1.0)}
    gather = None
    if dist is not None and world > 1 and not args.no_gather:
        gather = leg_gather(dist, world, rank, args.dist_backend)

    out = None
    if rank == 0:
        gates = n * world * steps_total
        value = gates / elapsed
        m_launch = n // chunks                       # gates per kernel launch
        ach = m_launch * ALG_BYTES_K3 / (k3_ms * 1e-3) / 1e9
        traffic, rocprof_ms, traffic_source, prof_file = None, None, None, None   # from the committed rocprofv3 passes of the same workload, see profiles/
        tf = os.path.join(ROOT, "profiles", "traffic_%s.json" % args.layout)
        if os.path.exists(tf) and m_launch == (1 << 20):
            prof = json.load(open(tf))
            kp = prof["k_beaver_finish_asm"]
            traffic = kp["hbm_bytes_per_launch"]
            rocprof_ms = kp.get("rocprof_avg_launch_ms")
            prof_file = "profiles/traffic_%s.json" % args.layout
            traffic_source = "profiles/traffic_%s.json: %s -- committed rocprofv3 PMC passes of this workload, NOT measured in this run" % (args.layout, prof.get("source", ""))
        ach_prof = (m_launch * ALG_BYTES_K3 / (rocprof_ms * 1e-3) / 1e9) if rocprof_ms else None
        if n * world == (1 << 24) and world > 1:
            wl = "2^24 AuthenticatedScalar Beaver muls over BN254 Fr sharded across %d GPUs, 2^%d contiguous gates per GPU per step (BASELINE.json configs[2])" % (world, args.log2n)
        else:
            wl = "2^%d AuthenticatedScalar Beaver muls over BN254 Fr per GPU per step, two parties in-process, mock net (BASELINE.json configs[1])" % args.log2n
        uniq = {json.dumps({k: v for k, v in i.items() if k not in ("pid", "local_device")}, sort_keys=True) + ("" if ("uuid" in i or "pci_bus_id" in i) else str(i["local_device"]))
                for i in idents}
        out = {
            "metric": METRIC, "value": value, "unit": "gates/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / steps_total * 1e3, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "u256 Montgomery (8 x u32 limbs, v_mad_u64_u32)", "data": "synthetic",
            "timed_rounds": rounds, "timed_steps_total": steps_total, "timed_region_ms": elapsed * 1e3,
            "ranks_seen": ranks_seen, "distinct_devices": len(uniq), "rank_devices": idents,
            "per_rank_ms_per_step": {"min": min(per_rank_ms), "max": max(per_rank_ms), "all": per_rank_ms},
            "config": {"workload": wl, "gates_per_gpu": n, "gates_per_step_all_gpus": n * world, "field": "bn254_fr", "layout": args.layout,
                       "launches_per_step": 4 * chunks, "gates_per_launch": m_launch, "workload_sets_rotated": len(sets),
                       "timed_region": "the K = %d steps run %d time(s) back to back between one pair of barriers (--min-timed-ms %.0f: a region of K steps alone would be "
                                       "%.1f ms); value and ms_per_step are over all %d steps" % (args.steps, rounds, args.min_timed_ms, est_region_ms, steps_total),
                       "settle_ms": args.settle_ms, "settle_note": "untimed run of the same pipeline before the warm-up steps, every rank: keeps the timed region out of the "
                                   "power controller's transient after idle (profiles/r02/ramp_probe.txt); --settle-ms 0 disables",
                       "headline_layout_note": "engine-native split columns (what gate outputs are kept in between gates; north_star allows SoA).  The arkworks AoS records the "
                                               "boundary receives run the same pipeline at the fraction reported as aos_pipeline_frac_of_hbm_peak",
                       "parallelism": "gate-range sharding, no data-path collective"},
            "roofline": {"bound": "hbm", "kernel": ("k_beaver_finish_asm_sw<0>" if args.layout == "split" else "k_beaver_finish_asm_aos<0>") + " (K2+K3 fused, hand-scheduled)",
                         "achieved": ach_prof if ach_prof else ach, "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": (ach_prof if ach_prof else ach) / HBM_PEAK_GBPS,
                         "frac_source": ("the committed rocprofv3 --kernel-trace average of this kernel on this workload (%s, rocprof_avg_launch_ms): the figure anyone can recompute "
                                         "from profiles/ -- it includes the profiler's own effect on the kernel; this run's own dispatch-bound HIP events give frac_hip_events" % prof_file) if ach_prof
                                        else "this run's dispatch-bound HIP events (no committed profile for this layout / launch size)",
                         "achieved_hip_events": ach, "frac_hip_events": ach / HBM_PEAK_GBPS,
                         "traffic": traffic, "traffic_source": traffic_source,
                         "algorithmic_bytes_per_launch": m_launch * ALG_BYTES_K3, "gates_per_launch": m_launch, "avg_launch_ms": k3_ms,
                         "avg_launch_ms_note": "HIP events bound to the kernel dispatch (hipExtLaunchKernelGGL) on sampled steps of the timed region",
                         "rocprof_avg_launch_ms": rocprof_ms,
                         "frac_rocprof": (ach_prof / HBM_PEAK_GBPS) if ach_prof else None,
                         "ceiling_note": "two-kernel pipeline: 580 B moved per 512 B counted per party-gate (the 64 B own-d||e re-read by K2+K3 and 4 B of K1 slack), so at the "
                                         "~6.3 TB/s the memory system sustains the pipeline tops out at 6.3 x 512/580 / 8 = 0.695 of the 8 TB/s peak (DESIGN.md section 3)",
                         "clock_effect": clock_effect()},
            "pipeline": {"algorithmic_GBps": n * ALG_BYTES_PER_GATE / (dev_ms_per_step * 1e-3) / 1e9,
                         "frac_of_hbm_peak": n * ALG_BYTES_PER_GATE / (dev_ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                         "k1_avg_launch_ms": k1_ms, "k3_avg_launch_ms": k3_ms, "device_ms_per_step": dev_ms_per_step, "steps_with_kernel_events": r["sampled"],
                         "k1_achieved_GBps": m_launch * ALG_BYTES_K1 / (k1_ms * 1e-3) / 1e9},
            "results_check": "open(batch_mul(x,y)) == x*y and MAC shares sum to key*x*y: %s" % ("ok" if ok else "FAILED"),
        }
        if rank_exact is not None:
            out["per_rank_oracle_check"] = rank_exact
        if cold is not None:
            out["value_cold"] = n * world * args.steps / cold["elapsed"]
            out["roofline"]["frac_cold"] = m_launch * ALG_BYTES_K3 / (cold["k3_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBPS
            out["cold"] = {"what": "the same %d warm-up + exactly %d timed steps (one round) run FIRST with no settle phase (--settle-ms 0): the GPU idled while the workload was built" % (args.warmup, args.steps),
                           "ms_per_step": cold["elapsed"] / args.steps * 1e3, "k1_avg_launch_ms": cold["k1_ms"], "k3_avg_launch_ms": cold["k3_ms"],
                           "device_ms_per_step": cold["dev_ms_per_step"]}
        if gather is not None:
            out["gather"] = gather
        if not args.no_cpu_baseline and world == 1:
            cb, res, myde, m = cpu_baseline(parties, n, args.cpu_log2n, args.layout)
            out["cpu_baseline"] = cb
            # the oracle's result for the same seeded workload is the bit-exact expectation for the buffers the timed region wrote
            exact = oracle_bitexact(parties, n, m, chunks, args.layout, res, myde)
            out["oracle_bitexact_gates"] = exact
            out["oracle_bitexact_note"] = "both parties' d||e and result records of workload set 0 after the timed region vs oracle/ark_oracle.c, every word, %d of %d gates compared" % (m, n)
            ok = ok and exact == m
        if world == 1 and not args.no_extras:
            del sets, parties, truth
            torch.cuda.empty_cache()
            out["end_to_end"], ok_e = leg_end_to_end(pkg, eng, dev, args.e2e_log2n)
            torch.cuda.empty_cache()
            out["circuit"], ok_c = leg_circuit(pkg, eng, dev, args.circuit_log2n, args.circuit_depth)
            torch.cuda.empty_cache()
            ok = ok and ok_c
            out["circuit_party_gates_per_s"] = out["circuit"]["party_gates_per_s"]
            out["circuit_frac_of_link_floor"] = out["circuit"]["frac_of_link_floor"]
            out["aos"], ok_a = leg_aos(eng, n, args)
            out["config4"], ok_4 = leg_config4(eng)
            torch.cuda.empty_cache()
            out["config5"], ok_5 = leg_config5(pkg, dev)
            ok = ok and ok_e and ok_a and ok_4 and ok_5
            # the figures a reader should not have to dig for, next to `value`
            out["aos_pipeline_frac_of_hbm_peak"] = out["aos"]["pipeline_frac_of_hbm_peak"]
            out["aos_gates_per_s"] = out["aos"]["gates_per_s"]
            out["end_to_end_party_gates_per_s"] = out["end_to_end"]["party_gates_per_s"]
            out["config5_end_to_end_ms"] = out["config5"]["end_to_end_ms"]
            out["config5_host_sha3_share_of_end_to_end"] = out["config5"].get("host_sha3_share_of_end_to_end")
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()
    if not ok:
        raise SystemExit("result check failed")


if __name__ == "__main__":
    main()
