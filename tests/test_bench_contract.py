"""The driver-facing contract of bench.py: rank 0 ends stdout with ONE compact JSON line (under 4 KB, numbers and short identifiers only)
with the agreed fields, for a plain launch and for a `python -m torch.distributed.run` launch (2 ranks; gloo backend here because the GPU box
has one GPU); every leg's full record goes to the detail file."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
            "data", "config", "roofline", "timed_rounds", "timed_steps_total", "timed_region_ms", "ranks_seen", "distinct_devices", "rank_devices",
            "per_rank_ms_per_step", "results_check", "pipeline_frac_of_hbm_peak"]
ROOFLINE_KEYS = {"bound", "kernel", "achieved", "peak", "unit", "frac", "frac_hip_events", "traffic", "algorithmic_bytes_per_launch", "gates_per_launch",
                 "avg_launch_ms", "rocprof_avg_launch_ms", "frac_priced_from"}
CPU_KEYS = {"value", "unit", "cores", "kind", "label", "sample", "single_thread_value", "fused_value", "fused_single_thread_value", "cpu_model", "nproc",
            "compiler", "flags", "cargo_probe"}
LINE_BUDGET = 4096


def _no_prose(o, path=""):
    """the compact line carries numbers and short identifiers: no string value longer than 160 characters, no nesting below two levels"""
    if isinstance(o, dict):
        assert path.count(".") < 2, "nested too deep: " + path
        for k, v in o.items():
            _no_prose(v, path + "." + k)
    elif isinstance(o, str):
        assert len(o) <= 160, "prose on the headline line at %s (%d chars)" % (path, len(o))


def _last_line(stdout):
    """the headline is the LAST stdout line, the only one, and under the budget"""
    lines = stdout.strip().splitlines()
    js = [ln for ln in lines if ln.startswith("{")]
    assert len(js) == 1 and lines[-1] == js[0], "stdout must hold exactly one JSON line, at its end"
    assert len(js[0]) < LINE_BUDGET, "headline line is %d bytes" % len(js[0])
    return js[0]


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _check(line, n_gpus, steps, warmup, scaling="weak"):
    d = json.loads(line)
    for k in REQUIRED:
        assert k in d, k
    assert d["n_gpus"] == n_gpus and d["steps"] == steps and d["warmup"] == warmup
    assert d["unit"] == "gates/s" and d["higher_is_better"] is True and d["scaling"] == scaling and d["vs_baseline"] is None
    # the timed region is whole rounds of the K steps, at least --min-timed-ms long, and the line says so
    assert d["timed_rounds"] >= 1 and d["timed_steps_total"] == steps * d["timed_rounds"]
    assert abs(d["ms_per_step"] * d["timed_steps_total"] - d["timed_region_ms"]) < 1e-6 * d["timed_region_ms"] + 1e-9
    assert d["timed_region_ms"] >= 50.0 or d["timed_rounds"] > 1 or steps * d["ms_per_step"] >= 50.0
    # self-proving rank / device census
    assert d["ranks_seen"] == n_gpus and len(d["rank_devices"]) == n_gpus and 1 <= d["distinct_devices"] <= n_gpus
    assert all(isinstance(i, str) and "@" in i for i in d["rank_devices"])
    pr = d["per_rank_ms_per_step"]
    assert pr["min"] <= pr["max"] and abs(pr["max"] - d["ms_per_step"]) < 0.5 * d["ms_per_step"]
    assert d["metric"].startswith("authenticated Beaver mul-gates/sec over BN254 Fr")
    assert "workload" in d["config"] and "model" not in d["config"] and {"gates_per_gpu", "field", "layout"} <= set(d["config"])
    r = d["roofline"]
    assert set(r) - {"frac_cold"} == ROOFLINE_KEYS
    _no_prose(d)
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and r["peak"] == 8000.0
    assert d["value"] > 0 and d["results_check"].endswith("ok")
    return d


def test_single_process_default_shape(tmp_path):
    det = str(tmp_path / "detail.json")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "6", "--warmup", "2", "--log2n", "16",
                        "--cpu-log2n", "12", "--circuit-log2n", "17", "--circuit-depth", "3", "--detail-file", det], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _check(_last_line(r.stdout), 1, 6, 2)
    cb = d["cpu_baseline"]
    assert set(cb) == CPU_KEYS
    assert cb["kind"] == "port" and cb["unit"] == "gates/s" and cb["cores"] >= 1 and cb["value"] > 0 and cb["sample"]
    # BASELINE.md section 3 in full: both CPU forms on all cores and on one thread, the host description, the cargo probe, the label
    assert cb["label"] == "CPU restatement of reference algorithm (not ark-mpc measured)"
    assert cb["fused_value"] > 0 and cb["fused_single_thread_value"] > 0 and cb["single_thread_value"] > 0
    assert "gcc" in cb["compiler"] and "-O3" in cb["flags"] and cb["cpu_model"] and cb["nproc"] >= 1 and cb["cargo_probe"]
    assert d["oracle_bitexact_gates"] == d["oracle_bitexact_of"] == 1 << 12
    # the top-level roofline fraction is the reproducible one; the live HIP-event figure rides beside it
    rf = d["roofline"]
    assert rf["frac_hip_events"] > 0 and rf["avg_launch_ms"] > 0
    # the N = 1 line leads with the figures a reader needs beside `value`: one scalar or two per leg, and whether each leg's check held
    for k in ("aos_pipeline_frac_of_hbm_peak", "aos_gates_per_s", "end_to_end_party_gates_per_s", "circuit_party_gates_per_s", "circuit_frac_of_link_floor",
              "config5_end_to_end_ms", "config4_ms"):
        assert d[k] > 0, k
    assert d["legs"] == {k: "ok" for k in ("end_to_end", "circuit", "aos", "config4", "config5")}
    assert 0 < d["circuit_frac_of_link_floor"] <= 1.05
    # ---- the detail file: every leg's full record ----
    assert d["detail_file"]
    D = json.load(open(det))
    assert D["headline"]["value"] == d["value"]
    fs = D["cpu_baseline"]["fused_single_pass"]
    assert fs["same_words_as_nine_passes"] is True and fs["value"] == cb["fused_value"]
    for k in ("cpu_model", "nproc", "compiler", "flags", "cargo_probe", "excludes"):
        assert D["cpu_baseline"].get(k), k
    assert "frac_source" in D["roofline"] and D["roofline"]["achieved_hip_events"] > 0
    # the circuit leg: resident operands, random triples from host memory, bit-exact, priced against the 192 B / party-gate link floor
    c = D["circuit"]
    assert c["results_check"].endswith("ok") and c["party_gates_per_s"] == d["circuit_party_gates_per_s"]
    assert set(c["modes"]) >= {"prefetched_async", "round4_blocking", "pageable_async", "sessions_resident_operands"}
    e2e = D["end_to_end"]
    for k in ("party_gates_per_s", "two_party_gates_per_s", "h2d_GBps", "d2h_GBps", "frac_of_measured_pcie"):
        assert k in e2e and e2e[k] is not None and e2e[k] > 0, k
    assert e2e["results_check"].endswith("ok") and e2e["party_gates_per_s"] == d["end_to_end_party_gates_per_s"]
    assert e2e["one_party"]["registered"]["path"] == {"phase1": "zero-copy kernel on the caller's vectors", "phase2": "zero-copy kernel on the caller's vectors"}
    assert e2e["one_party"]["pageable"]["path"] == {"phase1": "copy pipeline", "phase2": "copy pipeline"}
    assert e2e["wire_form"]["ms"] > 0 and e2e["wire_form"]["check"].endswith("ok")
    c4 = D["config4"]
    assert c4["host_vectors"]["check"].endswith("ok") and c4["host_vectors"]["pageable_ms"] > c4["ms"] and c4["ms"] == d["config4_ms"]
    assert 0 < c4["frac_of_nominal_valu_rate"] < c4["frac_of_int_alu_peak"] < 1
    assert D["config5"]["results_check"].endswith("ok") and D["aos"]["results_check"] == "ok"


def test_torchrun_two_ranks():
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1",
                        "--log2n", "16", "--dist-backend", "gloo"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _check(_last_line(r.stdout), 2, 4, 1)        # rank 0 only
    assert "cpu_baseline" not in d               # N = 1 only
    assert d["per_rank_oracle_check"]["ranks_all_exact"] is True and d["per_rank_oracle_check"]["gates_checked_per_rank"] == 4096
    assert d["distinct_devices"] == 1            # both ranks of this test sit on the box's one GPU, and the line says so


def test_torchrun_one_rank_over_rccl():
    """the driver's launcher with the backend it uses -- torch.distributed.run + nccl (= RCCL) -- with ONE rank on the box's one GPU: the process
    group is built on the device, and the barriers, the max / sum reductions, the object and tensor gathers of the headline all cross RCCL
    (world size 1; two ranks cannot share a GPU under RCCL, so the 2-rank tests above use gloo)"""
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "1",
                        "--log2n", "16", "--no-extras", "--cpu-log2n", "12"], capture_output=True, text=True, timeout=900, cwd=ROOT,
                       env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0, r.stderr[-2000:]
    d = _check(_last_line(r.stdout), 1, 4, 1)
    assert d["per_rank_oracle_check"]["ranks_all_exact"] is True and d["cpu_baseline"]["value"] > 0 and d["oracle_bitexact_gates"] == 1 << 12


def test_torchrun_two_ranks_strong_scaling():
    """--scaling strong: a fixed total (here 2^17 gates per step) cut into one contiguous range per rank"""
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1",
                        "--scaling", "strong", "--total-log2n", "17", "--dist-backend", "gloo"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _check(_last_line(r.stdout), 2, 4, 1, scaling="strong")
    assert d["config"]["gates_per_gpu"] == 1 << 16 and d["config"]["gates_per_step_all_gpus"] == 1 << 17


def _need_two_gpus():
    import torch
    n = torch.cuda.device_count()
    if n < 2:
        pytest.skip("needs 2 physical GPUs: torch.cuda.device_count() == %d on this box -- the cross-device legs (hipMemcpyPeerAsync between distinct devices, "
                    "peer mappings, RCCL with world > 1) have NOT been exercised here" % n)


def test_group_on_two_distinct_devices(tmp_path):
    """tests/c/group_oversub.c with members on devices 0 and 1: range kernels, peer pushes over xGMI, the pipelined commitment and the AND-reduced verify
    flag across two physical GPUs == one context"""
    _need_two_gpus()
    from test_abi_cpu import _build_c_smoke
    r = subprocess.run([_build_c_smoke(tmp_path, "group_oversub"), "100003", "2", "0", "1"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "group ok" in r.stdout, r.stdout + r.stderr


def test_config5_driver_on_two_distinct_devices():
    """tools/config5_dist.py over RCCL with one rank per physical GPU"""
    _need_two_gpus()
    env = dict(os.environ, LOG2N="20", DIST_BACKEND="nccl")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), os.path.join(ROOT, "tools", "config5_dist.py")], capture_output=True, text=True, timeout=900,
                       cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")][0])
    assert d["n_gpus"] == 2 and d["verify_ok"] is True and d["equals_single_slice_run"] is True


def test_bench_two_ranks_on_two_distinct_devices():
    """the driver's N = 2 launch on two physical GPUs over RCCL: the line must show 2 ranks on 2 distinct devices"""
    _need_two_gpus()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "10", "--warmup", "2"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _check(_last_line(r.stdout), 2, 10, 2)
    assert d["distinct_devices"] == 2 and d["per_rank_oracle_check"]["ranks_all_exact"] is True


def test_config5_driver_two_ranks_equals_single_slice():
    """tools/config5_dist.py with 2 ranks (gloo, both on the one GPU): sharded open + MAC check, ordered gather, one commitment;
    the gathered result must equal a single-slice run of the same code (uneven shards: n = 2^16 + 3 is not used, the driver takes
    powers of two, so the shards are even here; uneven gathers are covered by tests/test_sharding_gloo.py)."""
    env = dict(os.environ, LOG2N="16", DIST_BACKEND="gloo")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), os.path.join(ROOT, "tools", "config5_dist.py")], capture_output=True, text=True, timeout=900,
                       cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["verify_ok"] is True and d["equals_single_slice_run"] is True


def test_steps_above_2p20_gates_run_in_ranges():
    """--log2n 21 with the default --chunks 0: two 2^20-gate ranges per step (8 launches), results still verified."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "1", "--log2n", "21", "--no-cpu-baseline", "--no-extras"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(_last_line(r.stdout))
    assert d["config"]["launches_per_step"] == 8 and d["roofline"]["gates_per_launch"] == 1 << 20
    assert d["results_check"].endswith("ok") and d["value"] > 0


def test_single_process_group_end_to_end_leg():
    """`bench.py --single-process --only-e2e`: group sessions over members sharing device 0, per-member link rates and their sum in the line"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--single-process", "--gpus", "3", "--devices", "0,0,0", "--only-e2e", "--e2e-log2n", "16"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["members"] == 3 and d["oversubscribed"] is True and d["distinct_devices"] == 1 and d["results_check"].endswith("ok")
    assert len(d["per_member"]) == 3 and all(m["gates"] == 1 << 16 and m["session_link_up_GBps"] > 0 and m["phase1_link_up_GBps"] > 0 for m in d["per_member"])
    assert abs(sum(m["session_link_up_GBps"] for m in d["per_member"]) - d["link_up_GBps_sum_over_members"]) < 1e-6 * d["link_up_GBps_sum_over_members"]
    assert all(tuple(m["path"]) >= (1, 1) for m in d["per_member"])          # the members' phases ran in place on the pinned vectors
