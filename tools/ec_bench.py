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
asm = os.environ.get('ARKMPC_EC_ASM', '1') != '0'
limbs = 32 if os.environ.get("ARKMPC_EC_LIMBS") == "32" else 29
st = json.load(open(os.path.join(ROOT, "ark-mpc_amd", "csrc", "ec_asm_stats.json" if limbs == 32 else "ec29_asm_stats.json")))
per = st["mult_instrs_loop"] + st["mult_instrs_table"]
FQ_R01 = 27 * (5 * 7 + 2 * 16 + 1) + 8 * 7 + 7 * 16          # round 1's work definition: general multiplications per scalar-mul
res = {"workload": "2^%d PointShare x Scalar = %d scalar-muls" % (int(np.log2(n)), 2 * n), "ms": t * 1e3, "scalar_muls_per_s": 2 * n / t,
       "path": "hand-scheduled pipeline (digits, table, window loop, finish), %d-bit limbs" % limbs if asm else "compiled window loop (round 1)",
       "r01_accounting_frac_of_mad_only_peak": 2 * n * FQ_R01 * 136 / t / 31.2e12}
if asm:
    res.update({"mult_instrs_per_scalar_mul": per, "frac_of_int_alu_peak": 2 * n * per / t / 31.2e12})
print(json.dumps(res))
