"""The timed region of the headline: W warm-up + K timed steps of the two-party batch_mul pipeline between barriers, dispatch-bound HIP
events on sampled steps; and the word-for-word comparison of the timed buffers with the oracle."""
import json
import os
import time

import numpy as np
import torch

from .common import FID, ROOT, load_oracle, prepare_step, step


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


def clock_effect():
    """Measured effect of the profiler on the dominant kernel, from the committed PMC pass (profiles/r0N/clock_effect.json, written by
    tools/profile.sh): GRBM_GUI_ACTIVE cycles / the kernel's wall time under rocprofv3 = the shader clock it ran at while profiled."""
    for rnd in ("r06", "r05", "r04", "r03"):
        f = os.path.join(ROOT, "profiles", rnd, "clock_effect.json")
        if os.path.exists(f):
            d = json.load(open(f)); d["source"] = "profiles/%s/clock_effect.json" % rnd
            return d
    return None


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
    ora = load_oracle()
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

