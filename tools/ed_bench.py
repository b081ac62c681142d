#!/usr/bin/env python3
"""Curve25519 (Edwards) scalar-mul throughput: 2^18 PointShare x Scalar = 2^19 scalar-muls."""
import importlib, json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
pkg = importlib.import_module("ark-mpc_amd")
e = pkg.Engine("curve25519_fr", device=0, stream=torch.cuda.current_stream().cuda_stream)
n = 1 << int(os.environ.get("LOG2N", "18"))
g = torch.Generator(device="cuda"); g.manual_seed(7)
def rnd(cnt):
    raw = torch.randint(-(2**63), 2**63 - 1, (4 * cnt,), dtype=torch.int64, device="cuda", generator=g)
    out = torch.empty_like(raw); e.scalar_from_canonical(cnt, raw, out); return out
shares = torch.empty(32 * n, dtype=torch.int64, device="cuda")
e.scalarshare_mul_ed_generator(n, rnd(2 * n), shares)
sc = rnd(n); out = torch.empty_like(shares)
e.edshare_mul_public(n, shares, sc, out); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(3): e.edshare_mul_public(n, shares, sc, out)
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / 3 * 1e-3
print(json.dumps({"workload": "2^%d EdPointShare x Scalar = %d scalar-muls" % (int(np.log2(n)), 2 * n), "ms": t * 1e3, "scalar_muls_per_s": 2 * n / t}))
# generator multiplication (fixed-base table)
gs = rnd(2 * n)
e.scalarshare_mul_ed_generator(n, gs, shares); torch.cuda.synchronize()
e0.record()
for _ in range(3): e.scalarshare_mul_ed_generator(n, gs, shares)
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / 3 * 1e-3
print(json.dumps({"workload": "2^%d ScalarShare x Ed generator = %d generator muls" % (int(np.log2(n)), 2 * n), "fixed_base": os.environ.get("ARKMPC_NO_FIXED_BASE") != "1",
                  "ms": t * 1e3, "generator_muls_per_s": 2 * n / t}))
# variable-base MSM (bucket method on the complete Edwards addition) vs the per-element path it replaced (scalar-muls + a point sum)
import time
for lg in [int(x) for x in os.environ.get("MSM_LOG2N", "10,14,16,18,20").split(",")]:
    m = 1 << lg
    pts = torch.empty(16 * m, dtype=torch.int64, device="cuda")
    e.ed_generator_mul(m, rnd(m), pts)
    s = rnd(m); o = torch.empty(16, dtype=torch.int64, device="cuda")
    e.ed_msm(m, pts, s, o); torch.cuda.synchronize()
    reps = 3
    t0 = time.perf_counter()
    for _ in range(reps): e.ed_msm(m, pts, s, o)
    torch.cuda.synchronize(); t_msm = (time.perf_counter() - t0) / reps
    row = {"workload": "Curve25519 variable-base MSM, 2^%d points" % lg, "ms": t_msm * 1e3, "points_per_s": m / t_msm}
    if lg <= 18:
        tmp = torch.empty(16 * m, dtype=torch.int64, device="cuda")
        e.ed_scalar_mul(m, pts, s, tmp); e.ed_sum(m, tmp, o); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps): e.ed_scalar_mul(m, pts, s, tmp); e.ed_sum(m, tmp, o)
        torch.cuda.synchronize(); row["per_element_path_ms"] = (time.perf_counter() - t0) / reps * 1e3
    ss = rnd(2 * m); o2 = torch.empty(32, dtype=torch.int64, device="cuda")
    e.ed_msm_authenticated(m, pts, ss, o2); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): e.ed_msm_authenticated(m, pts, ss, o2)
    torch.cuda.synchronize(); row["authenticated_ms"] = (time.perf_counter() - t0) / reps * 1e3
    print(json.dumps(row))
