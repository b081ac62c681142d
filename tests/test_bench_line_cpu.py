"""bench.py's output discipline, checked without a GPU: emit() writes the detail file, prints exactly ONE stdout line, and that line stays
under the budget for an 8-rank record with every optional key present (the round-5 line outgrew the driver's record at 22 KB)."""
import argparse
import importlib.util
import io
import json
import os
import sys
from contextlib import redirect_stderr, redirect_stdout

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_entry", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _full_line(world=8):
    f = 1234567890.1234567
    line = {"metric": "authenticated Beaver mul-gates/sec over BN254 Fr, batch 2^20, at 1/2/4/8 GPUs", "value": f, "unit": "gates/s", "n_gpus": world, "steps": 200,
            "warmup": 20, "ms_per_step": f, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u256 (8 x u32 limbs, Montgomery, v_mad_u64_u32)",
            "data": "synthetic", "timed_rounds": 14, "timed_steps_total": 2800, "timed_region_ms": f, "ranks_seen": world, "distinct_devices": world,
            "rank_devices": ["%d@pci%d" % (i, 160 + i) for i in range(world)], "per_rank_ms_per_step": {"min": f, "max": f},
            "config": {"workload": "w" * 150, "gates_per_gpu": 1 << 21, "gates_per_step_all_gpus": 1 << 24, "field": "bn254_fr", "layout": "split",
                       "launches_per_step": 8, "settle_ms": 30.0, "parallelism": "gate-range sharding, no data-path collective"},
            "roofline": {k: f for k in ("achieved", "peak", "frac", "frac_hip_events", "traffic", "algorithmic_bytes_per_launch", "gates_per_launch", "avg_launch_ms",
                                        "rocprof_avg_launch_ms", "frac_cold")},
            "cpu_baseline": {k: f for k in ("value", "cores", "single_thread_value", "fused_value", "fused_single_thread_value", "nproc")},
            "pipeline_frac_of_hbm_peak": f, "k1_avg_launch_ms": f, "results_check": "open(batch_mul(x,y)) == x*y and MAC shares sum to key*x*y: ok",
            "per_rank_oracle_check": {"gates_checked_per_rank": 4096, "ranks_all_exact": True}, "value_cold": f, "gather_ms": f, "gather_bus_GBps_per_rank": f,
            "gather_ordered": True, "oracle_bitexact_gates": 1 << 20, "oracle_bitexact_of": 1 << 20, "legs": {k: "ok" for k in ("end_to_end", "circuit", "aos", "config4", "config5")}}
    line["roofline"].update({"bound": "hbm", "kernel": "k_beaver_finish_asm_sw<0> (K2+K3 fused)", "unit": "GB/s", "frac_priced_from": "profiles/traffic_split.json"})
    line["cpu_baseline"].update({"unit": "gates/s", "kind": "port", "label": "CPU restatement of reference algorithm (not ark-mpc measured)", "sample": "s" * 120,
                                 "cpu_model": "AMD EPYC 9575F 64-Core Processor", "compiler": "gcc (Ubuntu 11.4.0-1ubuntu1~22.04.2) 11.4.0", "flags": "-O3 -march=native -fPIC -pthread",
                                 "cargo_probe": "absent"})
    for k in ("aos_pipeline_frac_of_hbm_peak", "aos_gates_per_s", "end_to_end_party_gates_per_s", "end_to_end_two_party_gates_per_s", "end_to_end_frac_of_measured_pcie",
              "circuit_party_gates_per_s", "circuit_two_party_gates_per_s", "circuit_frac_of_link_floor", "config4_ms", "config4_scalar_muls_per_s", "config4_frac_of_int_alu_peak", "config5_end_to_end_ms",
              "config5_device_ms", "config5_host_sha3_share_of_end_to_end", "config5_device_frac_of_hbm_peak", "config5_split_device_frac_of_hbm_peak"):
        line[k] = f
    return line


def _emit(b, line, detail, path):
    args = argparse.Namespace(detail_file=path, legs_to_stderr=True)
    out, err = io.StringIO(), io.StringIO()
    with redirect_stdout(out), redirect_stderr(err):
        b.emit(line, detail, args)
    return out.getvalue(), err.getvalue()


def test_full_line_is_one_stdout_line_under_the_budget(tmp_path):
    b = _bench()
    det = str(tmp_path / "sub" / "detail.json")
    prose = {"what": "x" * 5000}
    out, err = _emit(b, _full_line(), {"end_to_end": prose, "circuit": prose, "roofline": {"frac_source": "y" * 900}}, det)
    lines = out.splitlines()
    assert len(lines) == 1 and len(lines[0]) < b.LINE_BUDGET, len(lines[0])
    d = json.loads(lines[0])
    assert d["detail_file"] and d["n_gpus"] == 8 and len(d["rank_devices"]) == 8
    D = json.load(open(det))
    assert D["end_to_end"] == prose and D["headline"]["value"] == d["value"]
    legs = [json.loads(ln) for ln in err.splitlines() if ln.startswith("{")]
    assert {l["leg"] for l in legs} == {"end_to_end", "circuit", "roofline"}


def test_line_never_exceeds_the_hard_cap_and_detail_failure_is_not_fatal(tmp_path):
    b = _bench()
    line = _full_line()
    line["rank_devices"] = ["r" * 900 for _ in range(8)]            # something went wrong upstream: optional keys are dropped, the contract keys stay
    out, err = _emit(b, line, {}, "/proc/nonexistent/detail.json")
    lines = out.splitlines()
    assert len(lines) == 1 and len(lines[0]) <= b.LINE_HARD_CAP
    d = json.loads(lines[0])
    assert "rank_devices" not in d and d["value"] > 0 and "roofline" in d and "cpu_baseline" in d and "detail_file" not in d
    assert "detail file not written" in err
