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
    ap.add_argument("--log2n", type=int, default=20, help="gates per GPU per step (default 2^20, the metric's batch)")
    ap.add_argument("--layout", choices=["aos", "split"], default="split",
                    help="HBM layout of share vectors: arkworks AoS (drop-in) or engine-native split columns")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-log2n", type=int, default=20, help="CPU baseline sample size (gates)")
    ap.add_argument("--no-check", action="store_true")
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


def build_workload(eng, n, seed, layout):
    gen = torch.Generator(device="cuda")
    gen.manual_seed(seed)
    key_sh = [rand_field_elems(eng, 1, gen), rand_field_elems(eng, 1, gen)]
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


def cpu_baseline(parties, n, log2n_cpu, layout):
    """The oracle's restatement of the reference's literal 9-pass batch_mul (both parties), timed on this host's
    cores on the first 2^log2n_cpu gates of the same workload (kind = "port"): all cores via a static range split
    in C (upper bound for the reference's rayon executor) and one thread (its default single executor thread)."""
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
    fn = ora.lib.ora_batch_mul_9pass_mt
    P = ora._p

    def timed(nthreads):
        t0 = time.perf_counter()
        for party in (0, 1):
            h = H[party]
            rc = fn(ctypes.c_int(FID), ctypes.c_size_t(m), ctypes.c_int(party), P(keys[party]), P(h["x"]), P(h["y"]), P(h["a"]), P(h["b"]),
                    P(h["c"]), P(de[1 - party]), P(myde[party]), P(res[party]), ctypes.c_int(nthreads))
            assert rc == 0
        return time.perf_counter() - t0

    timed(cores)  # warm
    reps, tot = 0, 0.0
    while tot < 6.0 and reps < 50:
        tot += timed(cores); reps += 1
    t_all = tot / reps
    reps1, tot1 = 0, 0.0
    while tot1 < 4.0 and reps1 < 10:
        tot1 += timed(1); reps1 += 1
    t_one = tot1 / reps1
    assert np.array_equal(myde[0], de[0]) and np.array_equal(myde[1], de[1])
    return {
        "value": m / t_all, "unit": "gates/s", "cores": cores, "kind": "port",
        "sample": "first 2^%d gates of the same seeded workload, both parties, the reference's literal 9-pass batch_mul "
                  "(oracle/ark_oracle.c ora_batch_mul_9pass_mt, gcc -O3), %d pthreads static range split, mean of %d runs; "
                  "single_thread_value = 1 thread, mean of %d runs" % (int(np.log2(m)), cores, reps, reps1),
        "single_thread_value": m / t_one,
    }, res, m


def main():
    args = parse()
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
    pkg = importlib.import_module("ark-mpc_amd")
    eng = pkg.Engine(FID, device=dev, host_buffers=False, stream=torch.cuda.current_stream().cuda_stream)
    n = 1 << args.log2n
    sets = [build_workload(eng, n, seed=0xA11CE002 + rank + 7919 * k, layout=args.layout) for k in range(max(1, args.sets))]
    parties, truth = sets[0]

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    if args.chunks <= 0:
        args.chunks = max(1, n >> 20)
    call_sets = [prepare_step(eng, n, ps, args.layout, args.chunks, args.k3_order) for ps, _ in sets]
    per_step = 4 * args.chunks                      # launches per step: per gate range K1(P0), K1(P1), K3(P0), K3(P1)
    for w in range(args.warmup):
        step(call_sets[w % len(call_sets)])
    barrier()
    # per-kernel durations: on sampled steps each of the four launches carries a dispatch-bound HIP event pair (64 slots)
    max_sampled = min(16, 64 // per_step)            # the engine has 64 kernel-timer slots
    every = max(1, args.event_every, -(-args.steps // max(1, max_sampled)))
    sampled = [s for s in range(args.steps) if s % every == 0][:max_sampled]
    slot_of = {s: per_step * i for i, s in enumerate(sampled)}
    ev_begin, ev_end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev_begin.record()
    for s in range(args.steps):
        step(call_sets[s % len(call_sets)], eng, slot_of.get(s))
    ev_end.record()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if sampled:
        seg = np.array([[eng.kernel_timer_ms(slot_of[s] + j) for j in range(per_step)] for s in sampled]).reshape(len(sampled), args.chunks, 4)  # ms
        k1_ms = float(seg[:, :, :2].mean())
        k3_ms = float(seg[:, :, 2:].mean())
    else:
        k1_ms = k3_ms = float("nan")
    dev_ms_per_step = ev_begin.elapsed_time(ev_end) / args.steps

    ok = True if args.no_check else all(check_results(eng, n, ps, tr, args.layout) for ps, tr in sets[:min(len(sets), args.steps)])

    out = None
    if rank == 0:
        gates = n * world * args.steps
        value = gates / elapsed
        m_launch = n // args.chunks                  # gates per kernel launch
        ach = m_launch * ALG_BYTES_K3 / (k3_ms * 1e-3) / 1e9
        traffic, rocprof_ms = None, None   # from the committed rocprofv3 passes of the same workload, see profiles/
        tf = os.path.join(ROOT, "profiles", "traffic_%s.json" % args.layout)
        if os.path.exists(tf) and args.log2n == 20 and args.chunks == 1:
            prof = json.load(open(tf))["k_beaver_finish_asm"]
            traffic = prof["hbm_bytes_per_launch"]
            rocprof_ms = prof.get("rocprof_avg_launch_ms")
        out = {
            "metric": METRIC, "value": value, "unit": "gates/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u256 Montgomery (8 x u32 limbs, v_mad_u64_u32)", "data": "synthetic",
            "config": {"workload": "2^%d AuthenticatedScalar Beaver muls over BN254 Fr per GPU per step, two parties in-process, "
                                   "mock net (BASELINE.json configs[1])" % args.log2n,
                       "gates_per_gpu": n, "field": "bn254_fr", "layout": args.layout, "launches_per_step": 4 * args.chunks,
                       "workload_sets_rotated": len(sets),
                       "parallelism": "gate-range sharding, no data-path collective"},
            "roofline": {"bound": "hbm", "kernel": ("k_beaver_finish_asm<0,NT>" if args.layout == "split" else "k_beaver_finish_asm_aos<0>") + " (K2+K3 fused, hand-scheduled)", "achieved": ach, "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": ach / HBM_PEAK_GBPS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": m_launch * ALG_BYTES_K3, "gates_per_launch": m_launch, "avg_launch_ms": k3_ms,
                         "avg_launch_ms_note": "HIP events bound to the kernel dispatch (hipExtLaunchKernelGGL) on sampled steps of the timed region",
                         "rocprof_avg_launch_ms": rocprof_ms},
            "pipeline": {"algorithmic_GBps": n * ALG_BYTES_PER_GATE / (dev_ms_per_step * 1e-3) / 1e9,
                         "frac_of_hbm_peak": n * ALG_BYTES_PER_GATE / (dev_ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                         "k1_avg_launch_ms": k1_ms, "k3_avg_launch_ms": k3_ms, "device_ms_per_step": dev_ms_per_step, "steps_with_kernel_events": len(sampled),
                         "k1_achieved_GBps": m_launch * ALG_BYTES_K1 / (k1_ms * 1e-3) / 1e9},
            "results_check": "open(batch_mul(x,y)) == x*y and MAC shares sum to key*x*y: %s" % ("ok" if ok else "FAILED"),
        }
        if not args.no_cpu_baseline and world == 1:
            cb, _, _ = cpu_baseline(parties, n, args.cpu_log2n, args.layout)
            out["cpu_baseline"] = cb
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()
    if not ok:
        raise SystemExit("result check failed")


if __name__ == "__main__":
    main()
