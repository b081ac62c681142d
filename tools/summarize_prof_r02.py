#!/usr/bin/env python3
"""Summarise tools/profile_r02.sh: per-kernel average durations from the kernel-trace stats, HBM bytes per launch from the
FETCH_SIZE / WRITE_SIZE passes (gfx950 corrections of MI355X_MICROARCH.md: KiB units, FETCH_SIZE doubled for wide coalesced
streams -- calibrated in round 1 on k_beaver_mask), VALU counters of the scalar-mul kernels.  Also writes traffic_<layout>.json."""
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


kernel_stats("trace_default", 24)
for layout in ("split", "aos"):
    avg = kernel_stats("trace_" + layout, 4)
    fetch, write = counter("pmc_fetch_" + layout, "FETCH_SIZE"), counter("pmc_write_" + layout, "WRITE_SIZE")
    res = {"source": "gpurun_out/prof_r02 (tools/profile_r02.sh): rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate runs of `bench.py --layout %s "
                     "--steps 100 --warmup 10 --no-cpu-baseline --no-extras`; KiB units; FETCH_SIZE doubled (gfx950 note in MI355X_MICROARCH.md, calibrated "
                     "on k_beaver_mask in round 1)" % layout, "workload": "bench.py --layout %s, 2^20 gates per launch" % layout}
    print("== HBM traffic per launch, layout %s" % layout)
    for name, tag in (("k_beaver_finish_asm", "k_beaver_finish_asm"), ("k_beaver_mask", "k_beaver_mask")):
        fk = [k for k in fetch if tag in k]
        if not fk:
            continue
        k = fk[0]
        fb, wb = fetch[k] * 1024 * 2.0, write.get(k, 0.0) * 1024
        ms = [v for kk, v in avg.items() if tag in kk]
        res[name] = {"fetch_bytes": fb, "write_bytes": wb, "hbm_bytes_per_launch": fb + wb, "rocprof_avg_launch_ms": (ms[0] / 1e6) if ms else None}
        print("  %-70s fetch %8.2f MB  write %8.2f MB  total %8.2f MB  (%.1f B per party-gate)  avg %s us" %
              (k[:70], fb / 1e6, wb / 1e6, (fb + wb) / 1e6, (fb + wb) / (1 << 20), ("%.2f" % (ms[0] / 1e3)) if ms else "?"))
    json.dump(res, open(os.path.join(out, "traffic_%s.json" % layout), "w"), indent=1)
kernel_stats("trace_ec", 8)
print("== scalar-mul kernels, PMC per launch (2^19 scalar-muls)")
names = ("SQ_WAVES", "SQ_INSTS_VALU", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE")
cs = {n: counter("pmc_ec", n) for n in names}
for k in sorted(cs["SQ_WAVES"], key=lambda k: -cs["SQ_INSTS_VALU"].get(k, 0))[:6]:
    if "smul" not in k and "scalar_mul" not in k:
        continue
    w = cs["SQ_WAVES"][k]
    print("  %-60s waves %6d  VALU/wave %9.0f  wave_cycles(quad)/wave %10.0f  issue-stall %4.1f%%  GUI_ACTIVE %10.0f" %
          (k[:60], w, cs["SQ_INSTS_VALU"][k] / w, cs["SQ_WAVE_CYCLES"][k] / w, 100 * cs["SQ_WAIT_INST_ANY"][k] / max(cs["SQ_WAVE_CYCLES"][k], 1), cs["GRBM_GUI_ACTIVE"][k]))
