"""AoS leg: the arkworks record layout exactly as the boundary receives it (64-byte ScalarShare records)."""
import json
import os

import torch

from .common import ALG_BYTES_PER_GATE, HBM_PEAK_GBPS, ROOT, build_workload, check_results, timed_events
from .pipeline import run_pipeline


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
    summary = {"aos_pipeline_frac_of_hbm_peak": out["pipeline_frac_of_hbm_peak"], "aos_gates_per_s": out["gates_per_s"]}
    return summary, out, ok and ok2

