"""Config 4 leg (BASELINE.json configs[3]): 2^18 PointShare x public Scalar over BN254 G1; integer-ALU bound."""
import ctypes
import importlib
import json
import os
import time

import numpy as np
import torch

from .common import FID, ROOT, load_oracle, rand_field_elems, timed_events

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
    ora = load_oracle()
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
    summary = {"config4_ms": ms, "config4_scalar_muls_per_s": smuls, "config4_frac_of_int_alu_peak": smuls * per_smul / MAD_PEAK_PER_S}
    return summary, {"workload": "2^18 PointShare x Scalar over BN254 G1 = 2^19 scalar-muls (BASELINE.json configs[3])", "ms": ms, "host_vectors": host,
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

