#!/usr/bin/env python3
"""BASELINE.json config 4 on one GPU: 2^18 PointShare x public Scalar (= 2^19 BN254 G1 scalar-muls), timed."""
import importlib, json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
pkg = importlib.import_module("ark-mpc_amd")
e = pkg.Engine(0, device=0, stream=torch.cuda.current_stream().cuda_stream)
n = 1 << int(os.environ.get("LOG2N", "18"))
g = torch.Generator(device="cuda"); g.manual_seed(0xA11CE004)
def rnd(cnt):
    raw = torch.randint(-(2**63), 2**63 - 1, (4 * cnt,), dtype=torch.int64, device="cuda", generator=g)
    out = torch.empty_like(raw); e.scalar_from_canonical(cnt, raw, out); return out
ss = rnd(2 * n)                       # n ScalarShares (discrete logs of the point shares)
shares = torch.empty(24 * n, dtype=torch.int64, device="cuda")
e.scalarshare_mul_generator(n, ss, shares)          # P_i = s_i G as PointShares
sc = rnd(n)
out = torch.empty_like(shares)
e.pointshare_mul_public(n, shares, sc, out); torch.cuda.synchronize()
reps = int(os.environ.get("REPS", "3"))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    e.pointshare_mul_public(n, shares, sc, out)
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / reps * 1e-3
glv = os.environ.get('ARKMPC_NO_GLV', '0') != '1'
# GLV: 33 windows x (4 dbl + 2 add + 1 beta mul) + table (7 dbl + 7 add); plain 4-bit windows: 256 dbl + 64 add + table
w4 = os.environ.get('ARKMPC_GLV_W', '5') == '4'
# Fq multiplications per scalar-mul: windows x (doublings + 2 additions + 1 beta mul) + table build
FQ_MULS = ((33 * (4 * 7 + 2 * 16 + 1) + 7 * 7 + 7 * 16) if w4 else (27 * (5 * 7 + 2 * 16 + 1) + 8 * 7 + 7 * 16)) if glv else (256 * 7 + 64 * 16 + 7 * 7 + 7 * 16)
print(json.dumps({"workload": "2^%d PointShare x Scalar = %d scalar-muls" % (int(np.log2(n)), 2 * n), "ms": t * 1e3,
                  "scalar_muls_per_s": 2 * n / t, "fq_muls_per_s": 2 * n * FQ_MULS / t,
                  "algorithm": ("glv + unsigned 4-bit windows" if w4 else "glv + signed 5-bit windows") if glv else "4-bit windows", "fq_muls_per_scalar_mul": FQ_MULS, "frac_of_mad_only_peak": 2 * n * FQ_MULS / t / (31.2e12 / 136)}))
