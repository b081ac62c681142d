#!/usr/bin/env python3
"""Variable-base MSM throughput (arkmpc_g1_msm / _authenticated) against the per-element path
(arkmpc_g1_scalar_mul + arkmpc_g1_sum).  Sizes 2^LOG2N (env, comma list)."""
import importlib, json, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
pkg = importlib.import_module("ark-mpc_amd")
e = pkg.Engine("bn254_fr", device=0, stream=torch.cuda.current_stream().cuda_stream)
g = torch.Generator(device="cuda"); g.manual_seed(11)
def rnd(cnt):
    raw = torch.randint(-(2**63), 2**63 - 1, (4 * cnt,), dtype=torch.int64, device="cuda", generator=g)
    out = torch.empty_like(raw); e.scalar_from_canonical(cnt, raw, out); return out
def timed(fn, reps):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for lg in [int(v) for v in os.environ.get("LOG2N", "10,14,18,20").split(",")]:
    n = 1 << lg
    pts = torch.empty(12 * n, dtype=torch.int64, device="cuda"); e.g1_generator_mul(n, rnd(n), pts)
    sc = rnd(2 * n)
    out = torch.empty(24, dtype=torch.int64, device="cuda")
    reps = 3 if lg >= 18 else 10
    t1 = timed(lambda: e.g1_msm(n, pts, sc, out), reps)
    t2 = timed(lambda: e.g1_msm_authenticated(n, pts, sc, out), reps)
    row = {"n": "2^%d" % lg, "msm_ms": round(t1, 3), "msm_points_per_s": round(n / t1 * 1e3), "msm_authenticated_ms": round(t2, 3),
           "authenticated_terms_per_s": round(2 * n / t2 * 1e3), "c": os.environ.get("ARKMPC_MSM_C", "auto")}
    if os.environ.get("AFFINE_BASES"):          # bases stored normalised (z = 1): the conversion pass is a copy
        xy = torch.empty(8 * n, dtype=torch.int64, device="cuda"); inf = torch.empty(n + 16, dtype=torch.uint8, device="cuda")
        e.g1_to_affine(n, pts, xy, inf)
        e3 = pkg.Engine("bn254_fq", device=0, stream=torch.cuda.current_stream().cuda_stream)
        onec = torch.zeros(4, dtype=torch.int64, device="cuda"); onec[0] = 1
        onem = torch.empty_like(onec); e3.scalar_from_canonical(1, onec, onem)
        pn = torch.cat([xy.view(n, 8), onem.view(1, 4).expand(n, 4)], dim=1).contiguous().view(-1)
        row["msm_normalised_bases_ms"] = round(timed(lambda: e.g1_msm(n, pn, sc, out), reps), 3)
    if lg <= 20 and not os.environ.get("SKIP_NAIVE"):
        tmp = torch.empty(12 * n, dtype=torch.int64, device="cuda")
        def naive():
            e.g1_scalar_mul(n, pts, sc, tmp); e.g1_sum(n, tmp, out)
        row["per_element_ms"] = round(timed(naive, 2), 3)
    print(json.dumps(row), flush=True)
