#!/usr/bin/env python3
"""Summarise tools/profile.sh (gpurun_out/prof_<round>): kernel stats, HBM bytes per launch (FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md),
the WRITE_SIZE calibration on pure store kernels with known byte counts, and the profiler's clock effect on the dominant kernel
(GRBM_GUI_ACTIVE cycles / kernel wall time under rocprofv3).  Writes traffic_split.json, clock_effect.json, write_calib.json."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def find(sub, pattern):
    r = glob.glob(os.path.join(out, sub, "**", pattern), recursive=True)
    return r[0] if r else None


def kernel_stats(sub, top=14):
    f = find(sub, "*kernel_stats.csv")
    rows = list(csv.DictReader(open(f))) if f else []
    print("== kernel stats: %s" % sub)
    for r in rows[:top]:
        print("  %-84s calls %6s  avg_us %10.2f  total_ms %9.3f  %5s%%" % (r.get("Name", "")[:84], r.get("Calls"), float(r.get("AverageNs", 0)) / 1e3,
                                                                           float(r.get("TotalDurationNs", 0)) / 1e6, r.get("Percentage")))
    return {r["Name"]: float(r["AverageNs"]) for r in rows}


def counter(sub, key):
    f = find(sub, "*counter_collection.csv")
    agg = defaultdict(lambda: [0.0, 0])
    if f:
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == key:
                agg[r.get("Kernel_Name", "")][0] += float(r.get("Counter_Value", 0)); agg[r.get("Kernel_Name", "")][1] += 1
    return {k: v[0] / max(v[1], 1) for k, v in agg.items()}


def trace_avg(sub):
    """average kernel duration (ns) per kernel name from a kernel_trace csv (passes that ran with --pmc + --kernel-trace)"""
    f = find(sub, "*kernel_trace.csv")
    agg = defaultdict(lambda: [0.0, 0])
    if f:
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"]][0] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"]); agg[r["Kernel_Name"]][1] += 1
    return {k: v[0] / max(v[1], 1) for k, v in agg.items()}


kernel_stats("trace_default", 24)
avg = kernel_stats("trace_split", 4)
fetch, write = counter("pmc_fetch_split", "FETCH_SIZE"), counter("pmc_write_split", "WRITE_SIZE")

# ---- WRITE_SIZE calibration: pure store kernels, known bytes
print("== WRITE_SIZE calibration (probes/write_calib, 2^21 elements): counter KiB * 1024 / bytes actually stored")
calib = {}
cw = counter("calib_write", "WRITE_SIZE")
plain = {}
pf = os.path.join(out, "write_calib_plain.jsonl")
if os.path.exists(pf):
    for line in open(pf):
        try:
            d = json.loads(line); plain[d["kernel"]] = d
        except Exception:
            pass
tags = {"pair": "k_pair<0>", "pair_nt": "k_pair<1>", "k3": "k_k3<0>", "k3_nt": "k_k3<1>", "line": "k_line<0>", "line_nt": "k_line<1>", "quad": "k_quad<0>", "quad_nt": "k_quad<1>"}
req = {n: counter("calib_wrreq", n) for n in ("TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_64B_sum")}
for name, tag in tags.items():
    ks = [k for k in cw if tag in k]
    if not ks or name not in plain:
        continue
    b = plain[name]["bytes_per_launch"]
    ratio = cw[ks[0]] * 1024 / b
    calib[name] = {"bytes_stored": b, "WRITE_SIZE_bytes": cw[ks[0]] * 1024, "ratio": ratio, "avg_us": plain[name]["avg_us"], "GBps": plain[name]["GBps_avg"]}
    extra = ""
    rq = [k for k in req["TCC_EA0_WRREQ_sum"] if tag in k]
    if rq:
        w, w64 = req["TCC_EA0_WRREQ_sum"][rq[0]], req["TCC_EA0_WRREQ_64B_sum"].get(rq[0], 0.0)
        calib[name]["WRREQ"] = w; calib[name]["WRREQ_64B"] = w64
        extra = "  WRREQ %.0f (64B: %.0f)" % (w, w64)
    print("  %-8s stored %9.2f MB  WRITE_SIZE %9.2f MB  ratio %.4f  %7.2f us  %7.1f GB/s%s" % (name, b / 1e6, cw[ks[0]] * 1024 / 1e6, ratio, plain[name]["avg_us"], plain[name]["GBps_avg"], extra))
json.dump(calib, open(os.path.join(out, "write_calib.json"), "w"), indent=1)

res = {"source": os.path.basename(out.rstrip("/")) + " (tools/profile.sh): rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate runs of `bench.py --layout split --steps 100 "
                 "--warmup 10 --no-cpu-baseline --no-extras --no-cold`; KiB units; FETCH_SIZE doubled (gfx950 note in MI355X_MICROARCH.md, calibrated on k_beaver_mask "
                 "in round 1); WRITE_SIZE checked on probes/write_calib (pure stores in the path's patterns, known byte counts): exact (ratio 1.0000) for plain stores, "
                 "+23..34 % REAL traffic for non-temporal 16-byte stores 32 B apart -- see write_calibration",
       "workload": "bench.py --layout split, 2^20 gates per launch",
       "write_calibration": calib if calib else "not re-run this round: profiles/r03/write_calib.json (probes/write_calib.hip)"}
print("== HBM traffic per launch, split layout")
for name, tag in (("k_beaver_finish_asm", "k_beaver_finish_asm"), ("k_beaver_mask", "k_beaver_mask")):
    fk = [k for k in fetch if tag in k]
    if not fk:
        continue
    k = fk[0]
    fb, wb = fetch[k] * 1024 * 2.0, write.get(k, 0.0) * 1024
    ms = [v for kk, v in avg.items() if tag in kk]
    # WRITE_SIZE needs no correction: on plain stores in every pattern of the path it equals the bytes stored (ratio 1.0000 above); what it
    # shows ABOVE the stored bytes for non-temporal stores is real write traffic (the same kernels run 1.3-1.9x longer)
    res[name] = {"fetch_bytes": fb, "write_bytes": wb, "hbm_bytes_per_launch": fb + wb, "rocprof_avg_launch_ms": (ms[0] / 1e6) if ms else None}
    print("  %-60s fetch %8.2f MB  write %8.2f MB  total %8.2f MB  (%.1f B per party-gate)  avg %s us" %
          (k[:60], fb / 1e6, wb / 1e6, (fb + wb) / 1e6, (fb + wb) / (1 << 20), ("%.2f" % (ms[0] / 1e3)) if ms else "?"))
json.dump(res, open(os.path.join(out, "traffic_split.json"), "w"), indent=1)
# the same for the arkworks AoS layout
avg_a = kernel_stats("trace_aos", 4)
fetch_a, write_a = counter("pmc_fetch_aos", "FETCH_SIZE"), counter("pmc_write_aos", "WRITE_SIZE")
res_a = {"source": os.path.basename(out.rstrip("/")) + " (tools/profile.sh): the same passes with --layout aos", "workload": "bench.py --layout aos, 2^20 gates per launch"}
print("== HBM traffic per launch, AoS layout")
for name, tag in (("k_beaver_finish_asm", "k_beaver_finish_asm"), ("k_beaver_mask", "k_beaver_mask")):
    fk = [k for k in fetch_a if tag in k]
    if not fk:
        continue
    k = fk[0]
    fb, wb = fetch_a[k] * 1024 * 2.0, write_a.get(k, 0.0) * 1024
    ms = [v for kk, v in avg_a.items() if tag in kk]
    res_a[name] = {"fetch_bytes": fb, "write_bytes": wb, "hbm_bytes_per_launch": fb + wb, "rocprof_avg_launch_ms": (ms[0] / 1e6) if ms else None}
    print("  %-60s fetch %8.2f MB  write %8.2f MB  total %8.2f MB  (%.1f B per party-gate)  avg %s us" %
          (k[:60], fb / 1e6, wb / 1e6, (fb + wb) / 1e6, (fb + wb) / (1 << 20), ("%.2f" % (ms[0] / 1e3)) if ms else "?"))
if len(res_a) > 2:
    json.dump(res_a, open(os.path.join(out, "traffic_aos.json"), "w"), indent=1)

# ---- clock effect: GRBM_GUI_ACTIVE / wall under the profiler
print("== clock under the profiler: GRBM_GUI_ACTIVE cycles per launch / kernel wall time (rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace)")
gui = counter("pmc_clock", "GRBM_GUI_ACTIVE")
tavg = trace_avg("pmc_clock")
ce = {}
bench_line = None
try:
    bench_line = json.loads(open(os.path.join(out, "bench_default_run.json")).read().strip().splitlines()[-1])
except Exception:
    pass
for tag in ("k_beaver_finish_asm", "k_beaver_mask"):
    ks = [k for k in gui if tag in k]
    if not ks or ks[0] not in tavg:
        continue
    cyc, ns = gui[ks[0]], tavg[ks[0]]
    un = [v for kk, v in avg.items() if tag in kk]
    ce[tag] = {"GRBM_GUI_ACTIVE_cycles_per_launch_sum_of_8_XCDs": cyc, "kernel_wall_us_in_pmc_pass": ns / 1e3, "busy_MHz_per_XCD_in_pmc_pass": cyc / 8 / ns * 1e3,
               "kernel_wall_us_kernel_trace_only": (un[0] / 1e3) if un else None}
    print("  %-24s %10.0f cycles (8 XCDs)  %8.2f us (PMC pass)  -> %7.1f busy-MHz per XCD ; kernel-trace-only pass %s us" % (tag, cyc, ns / 1e3, cyc / 8 / ns * 1e3,
          ("%.2f" % (un[0] / 1e3)) if un else "?"))
if bench_line:
    ev = bench_line["roofline"]["avg_launch_ms"] * 1e3
    ce["hip_event_us_unprofiled_run"] = ev
    if "k_beaver_finish_asm" in ce and ce["k_beaver_finish_asm"]["kernel_wall_us_kernel_trace_only"]:
        tr = ce["k_beaver_finish_asm"]["kernel_wall_us_kernel_trace_only"]
        ce["rocprof_over_hip_event"] = tr / ev
        print("  K2+K3: un-profiled HIP events %.2f us, rocprofv3 kernel trace %.2f us: ratio %.3f" % (ev, tr, tr / ev))
ce["note"] = ("GRBM_GUI_ACTIVE counts GPU-busy cycles at the shader clock, summed over the 8 XCDs, and includes the dispatch ramp around a kernel: for 37-60 us kernels "
              "busy cycles / kernel wall time overshoots the 2.4 GHz engine clock (it is an upper bound), for the 6.6 ms k_g1_smul_loop it reads 2.16 GHz (profiles/r03/summary.txt). "
              "What the profiler costs is read directly: the same kernel's average under rocprofv3 --kernel-trace over its un-profiled dispatch-bound HIP-event duration "
              "(rocprof_over_hip_event, 1.03); the bench line's `frac` uses the HIP events, `frac_rocprof` the kernel-trace average")
json.dump(ce, open(os.path.join(out, "clock_effect.json"), "w"), indent=1)
print("== single-process group (members sharing device 0)")
sp = os.path.join(out, "bench_single_process.jsonl")
if os.path.exists(sp):
    for line in open(sp):
        try:
            d = json.loads(line)
            print("  members %d (distinct devices %d): %.3e gates/s, %.3f ms/step, gather %s" % (d["ranks_seen"], d["distinct_devices"], d["value"], d["ms_per_step"],
                  json.dumps({k: round(v["GBps"], 1) for k, v in d.get("gather", {}).items() if isinstance(v, dict)})))
        except Exception as ex:
            print("  unparsable line: %r" % ex)

print("== VALU issue rate of the hand-scheduled kernels (per launch; SQ_INSTS_VALU wave-instructions, GRBM_GUI_ACTIVE summed over the 8 XCDs)")
print("   SIMD cycles per VALU instruction = 1024 SIMDs x busy cycles per XCD / (waves x VALU per wave).  A wave64 instruction occupies a SIMD16 for 4 cycles: 4.0 is the issue limit,")
print("   whatever the instruction (the round-1 cost model, 5.04 cycles per v_mad_u64_u32 and 2.8 per other VALU instruction, over-predicts these streams and is superseded);")
print("   a denser multiplier mix shows up as a LOWER CLOCK (busy cycles / wall time), not as more cycles per instruction")
import json as _json
def _stats(name):
    try:
        return _json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ark-mpc_amd", "csrc", name)))
    except Exception:
        return {}
for sub, tags in (("pmc_ec", ("k_g1_smul_loop", "k_g1_smul_table")), ("pmc_k3", ("k_beaver_finish_asm", "k_beaver_mask"))):
    cs = {n: counter(sub, n) for n in ("SQ_WAVES", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_WAIT_INST_ANY")}
    for tag in tags:
        ks = [k for k in cs["SQ_WAVES"] if tag in k]
        if not ks:
            continue
        k = ks[0]
        name = k.split("(")[0]
        st = _stats("ec29_asm_stats.json" if name.endswith("29") else "ec_asm_stats.json")
        w, valu, gui = cs["SQ_WAVES"][k], cs["SQ_INSTS_VALU"][k], cs["GRBM_GUI_ACTIVE"][k]
        per_wave = valu / max(w, 1)
        mult = {"k_g1_smul_loop": st.get("mult_instrs_loop"), "k_g1_smul_table": st.get("mult_instrs_table"), "k_beaver_finish_asm": 536, "k_beaver_mask": 0}.get(tag)
        line = "  %-24s waves %6d  VALU/wave %9.0f  busy cycles/XCD %11.0f  SIMD cycles per VALU instruction %.2f" % (
            name, w, per_wave, gui / 8, 1024 * (gui / 8) / max(w * per_wave, 1))
        if mult:
            line += "  | multiplier instructions/wave %d = %.2f of the stream" % (mult, mult / per_wave)
        print(line)

print("== streaming host-to-host sessions (bench.py --only-e2e), un-profiled run and copy / kernel trace")
for fn in ("bench_e2e_run.json",):
    try:
        d = json.loads(open(os.path.join(out, fn)).read().strip().splitlines()[-1])
        for k in ("pageable", "registered"):
            r = d["one_party"][k]
            print("  one party, %-10s: %.2f ms per 2^20 gates = %.3e party-gates/s, up %.1f GB/s, down %.1f GB/s, %.2f of the measured link (%.1f GB/s); phases: %s | %s" %
                  (k, r["ms"], r["party_gates_per_s"], r["h2d_GBps"], r["d2h_GBps"], r["frac_of_measured_pcie"], d["measured_pcie"]["h2d_GBps"],
                   r.get("path", {}).get("phase1", "?"), r.get("path", {}).get("phase2", "?")))
        t = d["two_party_one_gpu"]
        print("  two parties on one GPU / one link, one host thread driving both sessions: %.2f ms = %.3e two-party gates/s (a thread + context per party: %.2f ms)" %
              (t["ms"], t["two_party_gates_per_s"], t.get("two_host_threads_two_contexts", {}).get("ms", float("nan"))))
        print("  " + d["results_check"])
    except Exception as ex:
        print("  (no e2e run: %r)" % ex)
try:
    tr = json.load(open(os.path.join(out, "e2e_trace", "summary.json")))
    print("  trace of one copy-pipeline session (pageable vectors) under rocprofv3: " + json.dumps(tr["last_one_party_session"]))
    print("  zero-copy sessions (pinned vectors) under rocprofv3, medians: " + json.dumps(tr.get("zero_copy_one_party_sessions", {}).get("median")))
except Exception as ex:
    print("  (no e2e trace summary: %r)" % ex)
