"""The headline measurement: BASELINE.json configs[1] (2^20 AuthenticatedScalar Beaver muls over BN254 Fr per GPU per step, both parties
in-process, mock net), one process per GPU.  Returns the compact headline record (what bench.py prints) and the full one (detail file)."""
import json
import os

import numpy as np
import torch

from .common import ALG_BYTES_K1, ALG_BYTES_K3, ALG_BYTES_PER_GATE, HBM_PEAK_GBPS, METRIC, ROOT, build_workload, check_results, load_oracle
from .pipeline import clock_effect, oracle_bitexact, per_rank_oracle_check, rank_identity, run_pipeline

DTYPE = "u256 (8 x u32 limbs, Montgomery, v_mad_u64_u32)"


class Ranks:
    """the process group as the headline needs it: barrier, max / sum over ranks, gathers (all no-ops for a plain single-process launch)"""

    def __init__(self, dist, world, rank, local_rank, backend):
        self.dist, self.world, self.rank, self.local_rank = dist, world, rank, local_rank
        self.cdev = "cuda" if backend == "nccl" else "cpu"       # where collective tensors live

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()
        if torch.cuda.is_available():          # (always, where bench.py runs; the class itself is also exercised on CPU with gloo: tests/test_sharding_gloo.py)
            torch.cuda.synchronize()

    def reduce(self, v, op):
        if self.dist is None:
            return v
        t = torch.tensor([v], dtype=torch.float64, device=self.cdev)
        self.dist.all_reduce(t, op=getattr(self.dist.ReduceOp, op))
        return float(t.item())

    def gather_floats(self, v):
        if self.dist is None:
            return [v]
        t = torch.tensor([v], dtype=torch.float64, device=self.cdev)
        allt = [torch.zeros_like(t) for _ in range(self.world)]
        self.dist.all_gather(allt, t)
        return [float(x.item()) for x in allt]

    def gather_objects(self, o):
        if self.dist is None:
            return [o]
        out = [None] * self.world
        self.dist.all_gather_object(out, o)
        return out


def short_device(i):
    """one rank's GPU as a short string: local id @ PCI bus (or the head of the UUID)"""
    if "pci_bus_id" in i:
        return "%d@pci%s" % (i["local_device"], i["pci_bus_id"])
    if "uuid" in i:
        return "%d@%s" % (i["local_device"], str(i["uuid"])[:13])
    return "%d@pid%d" % (i["local_device"], i["pid"])


def committed_profile(layout, m_launch):
    """the committed rocprofv3 passes of this workload (profiles/traffic_<layout>.json, written by tools/profile.sh + tools/summarize_prof.py):
    HBM bytes per launch from the PMC counters and the kernel-trace average launch time.  Only for the 2^20-gate launch they were taken on."""
    tf = os.path.join(ROOT, "profiles", "traffic_%s.json" % layout)
    if not (os.path.exists(tf) and m_launch == (1 << 20)):
        return None
    prof = json.load(open(tf))
    kp = prof["k_beaver_finish_asm"]
    return {"traffic": kp["hbm_bytes_per_launch"], "rocprof_ms": kp.get("rocprof_avg_launch_ms"), "file": "profiles/traffic_%s.json" % layout,
            "source": prof.get("source", "")}


def run_headline(args, eng, ranks, n):
    """-> (line, detail, ok, sets, chunks) ; line / detail are None on ranks other than 0"""
    world, rank = ranks.world, ranks.rank
    dev = torch.cuda.current_device()
    sets = [build_workload(eng, n, seed=0xA11CE002 + rank + 7919 * k, layout=args.layout) for k in range(max(1, args.sets))]
    parties, truth = sets[0]
    # who is here: every rank adds 1 over the collective backend (RCCL under the driver) and contributes the identity of its GPU
    ranks_seen = int(round(ranks.reduce(1.0, "SUM")))
    idents = ranks.gather_objects(rank_identity(dev))
    # cold pass FIRST: exactly W warm-up + K timed steps with no settle phase -- what a K-step region measures on a GPU that was idle while
    # the workload was built (the power controller's transient, profiles/r02/ramp_probe.txt).  Reported beside the headline; its duration
    # also sizes the number of rounds of the headline region.
    cold = None
    if not args.no_cold and args.settle_ms > 0:
        rc_ = run_pipeline(eng, n, sets, args.layout, args, args.steps, args.warmup, ranks.barrier, settle_ms=0)
        cold = {"elapsed": ranks.reduce(rc_["elapsed"], "MAX"), "k3_ms": rc_["k3_ms"], "k1_ms": rc_["k1_ms"], "dev_ms_per_step": rc_["dev_ms_per_step"]}
        est_region_ms = cold["elapsed"] * 1e3
    else:
        rc_ = run_pipeline(eng, n, sets, args.layout, args, min(8, args.steps), 2, ranks.barrier, settle_ms=0)
        est_region_ms = ranks.reduce(rc_["elapsed"], "MAX") * 1e3 * args.steps / min(8, args.steps)
    rounds = 1
    if args.min_timed_ms > 0 and est_region_ms > 0:
        rounds = max(1, int(np.ceil(1.2 * args.min_timed_ms / est_region_ms)))     # (the sizing pass runs cold and a little slow: 20 % margin)
    rounds = int(ranks.reduce(float(rounds), "MAX"))       # one number on every rank
    r = run_pipeline(eng, n, sets, args.layout, args, args.steps, args.warmup, ranks.barrier, rounds=rounds)
    k1_ms, k3_ms, dev_ms_per_step, chunks = r["k1_ms"], r["k3_ms"], r["dev_ms_per_step"], r["chunks"]
    elapsed = ranks.reduce(r["elapsed"], "MAX")
    steps_total = args.steps * rounds
    per_rank_ms = ranks.gather_floats(r["elapsed"] / steps_total * 1e3)

    ok = True if args.no_check else all(check_results(eng, n, ps, tr, args.layout) for ps, tr in sets[:min(len(sets), args.steps)])
    # parity on every rank (N = 1 runs the full-size comparison in bench.py's cpu_baseline step instead)
    rank_exact = None
    if ranks.dist is not None and not args.no_check:
        if ranks.local_rank == 0:      # the checker library is (re)built by `make` on first load: one rank of the node does that, the others load the finished file
            load_oracle()
        ranks.dist.barrier()
        exact_r, m_r = per_rank_oracle_check(parties, n, chunks, args.layout)
        ok = ok and exact_r == m_r
        rank_exact = {"gates_checked_per_rank": m_r, "ranks_all_exact": bool(ranks.reduce(1.0 if exact_r == m_r else 0.0, "MIN") == 1.0)}
    if rank != 0:
        return None, None, ok, sets, chunks

    gates = n * world * steps_total
    m_launch = n // chunks                       # gates per kernel launch
    ach = m_launch * ALG_BYTES_K3 / (k3_ms * 1e-3) / 1e9
    prof = committed_profile(args.layout, m_launch)
    rocprof_ms = prof["rocprof_ms"] if prof else None
    ach_prof = (m_launch * ALG_BYTES_K3 / (rocprof_ms * 1e-3) / 1e9) if rocprof_ms else None
    if n * world == (1 << 24) and world > 1:
        wl = "2^24 AuthenticatedScalar Beaver muls over BN254 Fr sharded across %d GPUs, 2^%d contiguous gates per GPU per step (BASELINE.json configs[2])" % (world, args.log2n)
    else:
        wl = "2^%d AuthenticatedScalar Beaver muls over BN254 Fr per GPU per step, two parties in-process, mock net (BASELINE.json configs[1])" % args.log2n
    uniq = {json.dumps({k: v for k, v in i.items() if k not in ("pid", "local_device")}, sort_keys=True) + ("" if ("uuid" in i or "pci_bus_id" in i) else str(i["local_device"]))
            for i in idents}
    kernel = ("k_beaver_finish_asm_sw<0>" if args.layout == "split" else "k_beaver_finish_asm_aos<0>") + " (K2+K3 fused)"
    frac = (ach_prof if ach_prof else ach) / HBM_PEAK_GBPS
    # ---- the compact record: what bench.py prints as its ONE stdout line (numbers and short identifiers, no prose; budget 4 KB) ----
    line = {
        "metric": METRIC, "value": gates / elapsed, "unit": "gates/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / steps_total * 1e3, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
        "dtype": DTYPE, "data": "synthetic",
        "timed_rounds": rounds, "timed_steps_total": steps_total, "timed_region_ms": elapsed * 1e3,
        "ranks_seen": ranks_seen, "distinct_devices": len(uniq), "rank_devices": [short_device(i) for i in idents],
        "per_rank_ms_per_step": {"min": min(per_rank_ms), "max": max(per_rank_ms)},
        "config": {"workload": wl, "gates_per_gpu": n, "gates_per_step_all_gpus": n * world, "field": "bn254_fr", "layout": args.layout,
                   "launches_per_step": 4 * chunks, "settle_ms": args.settle_ms, "parallelism": "gate-range sharding, no data-path collective"},
        "roofline": {"bound": "hbm", "kernel": kernel, "achieved": ach_prof if ach_prof else ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": frac,
                     "frac_hip_events": ach / HBM_PEAK_GBPS, "traffic": prof["traffic"] if prof else None,
                     "algorithmic_bytes_per_launch": m_launch * ALG_BYTES_K3, "gates_per_launch": m_launch, "avg_launch_ms": k3_ms,
                     "rocprof_avg_launch_ms": rocprof_ms, "frac_priced_from": prof["file"] if ach_prof else "hip_events"},
        "pipeline_frac_of_hbm_peak": n * ALG_BYTES_PER_GATE / (dev_ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBPS,
        "k1_avg_launch_ms": k1_ms,
        "results_check": "open(batch_mul(x,y)) == x*y and MAC shares sum to key*x*y: %s" % ("ok" if ok else "FAILED"),
    }
    if rank_exact is not None:
        line["per_rank_oracle_check"] = rank_exact
    if cold is not None:
        line["value_cold"] = n * world * args.steps / cold["elapsed"]
    # ---- everything else: the detail file ----
    detail = {
        "rank_devices": idents, "per_rank_ms_per_step": {"min": min(per_rank_ms), "max": max(per_rank_ms), "all": per_rank_ms},
        "config": {"workload_sets_rotated": len(sets), "gates_per_launch": m_launch,
                   "timed_region": "the K = %d steps run %d time(s) back to back between one pair of barriers (--min-timed-ms %.0f: a region of K steps alone would be "
                                   "%.1f ms); value and ms_per_step are over all %d steps" % (args.steps, rounds, args.min_timed_ms, est_region_ms, steps_total),
                   "settle_note": "untimed run of the same pipeline before the warm-up steps, every rank: keeps the timed region out of the "
                                  "power controller's transient after idle (profiles/r02/ramp_probe.txt); --settle-ms 0 disables",
                   "headline_layout_note": "engine-native split columns (what gate outputs are kept in between gates; north_star allows SoA).  The arkworks AoS records the "
                                           "boundary receives run the same pipeline at the fraction reported as aos_pipeline_frac_of_hbm_peak"},
        "roofline": {"frac_source": ("the committed rocprofv3 --kernel-trace average of this kernel on this workload (%s, rocprof_avg_launch_ms): the figure anyone can recompute "
                                     "from profiles/ -- it includes the profiler's own effect on the kernel; this run's own dispatch-bound HIP events give frac_hip_events" % prof["file"]) if ach_prof
                                    else "this run's dispatch-bound HIP events (no committed profile for this layout / launch size)",
                     "achieved_hip_events": ach, "frac_rocprof": (ach_prof / HBM_PEAK_GBPS) if ach_prof else None,
                     "traffic_source": ("%s: %s -- committed rocprofv3 PMC passes of this workload, NOT measured in this run" % (prof["file"], prof["source"])) if prof else None,
                     "avg_launch_ms_note": "HIP events bound to the kernel dispatch (hipExtLaunchKernelGGL) on sampled steps of the timed region",
                     "ceiling_note": "two-kernel pipeline: 580 B moved per 512 B counted per party-gate (the 64 B own-d||e re-read by K2+K3 and 4 B of K1 slack), so at the "
                                     "~6.3 TB/s the memory system sustains the pipeline tops out at 6.3 x 512/580 / 8 = 0.695 of the 8 TB/s peak (DESIGN.md section 3)",
                     "clock_effect": clock_effect()},
        "pipeline": {"algorithmic_GBps": n * ALG_BYTES_PER_GATE / (dev_ms_per_step * 1e-3) / 1e9,
                     "frac_of_hbm_peak": n * ALG_BYTES_PER_GATE / (dev_ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                     "k1_avg_launch_ms": k1_ms, "k3_avg_launch_ms": k3_ms, "device_ms_per_step": dev_ms_per_step, "steps_with_kernel_events": r["sampled"],
                     "k1_achieved_GBps": m_launch * ALG_BYTES_K1 / (k1_ms * 1e-3) / 1e9},
    }
    if cold is not None:
        line["roofline"]["frac_cold"] = m_launch * ALG_BYTES_K3 / (cold["k3_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBPS
        detail["cold"] = {"what": "the same %d warm-up + exactly %d timed steps (one round) run FIRST with no settle phase (--settle-ms 0): the GPU idled while the workload was built" % (args.warmup, args.steps),
                          "ms_per_step": cold["elapsed"] / args.steps * 1e3, "k1_avg_launch_ms": cold["k1_ms"], "k3_avg_launch_ms": cold["k3_ms"],
                          "device_ms_per_step": cold["dev_ms_per_step"]}
    return line, detail, ok, sets, chunks
