"""cpu_baseline leg: the oracle's restatement of the reference's batch_mul timed on this host's cores."""
import ctypes
import os
import time

import numpy as np

from .common import FID, ROOT, host_cores, load_oracle


def host_description():
    """nproc, CPU model, compiler and flags beside the CPU figure; step 1: the cargo probe"""
    import shutil, subprocess
    model = None
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.lower().startswith("model name"):
                model = ln.split(":", 1)[1].strip(); break
    except OSError:
        pass
    def first_line(cmd):
        try:
            return subprocess.run(cmd, capture_output=True, text=True, timeout=20).stdout.strip().splitlines()[0]
        except Exception:            # noqa: BLE001
            return None
    flags = None
    try:
        for ln in open(os.path.join(ROOT, "oracle", "Makefile")):
            if ln.startswith("CFLAGS"):
                flags = ln.split("=", 1)[1].strip(); break
    except OSError:
        pass
    cargo = shutil.which("cargo")
    return {"cpu_model": model, "nproc": os.cpu_count(), "compiler": first_line([os.environ.get("CC", "gcc"), "--version"]), "flags": flags,
            "cargo_probe": (first_line(["cargo", "--version"]) or "present but not runnable") if cargo else "absent (`cargo` not on PATH): the reference's own "
                           "`cargo bench --bench batch_ops` cannot run on this box; the CPU restatement below is timed instead (BASELINE.md section 3 steps 1-2)"}


def cpu_baseline(parties, n, log2n_cpu, layout):
    """the oracle's restatement of the reference's batch_mul (both parties) timed on this host's cores on the first
    2^log2n_cpu gates of the same workload (kind = "port"), in BOTH forms the plan names -- the literal nine passes (authenticated_scalar.rs:
    848-879) and the fused single pass (:799-843 per element), so that the comparison is not hobbled by pass count -- each on all cores (static
    range split in C: the upper bound for the reference's rayon executor) and on one thread (its default single executor thread)."""
    ora = load_oracle()
    m = min(n, 1 << log2n_cpu)

    def host_aos(t):
        if layout == "aos":
            return t[:8 * m].cpu().numpy().view(np.uint64).copy()
        s = t[:4 * m].cpu().numpy().view(np.uint64).reshape(m, 4)
        mm = t[4 * n:4 * n + 4 * m].cpu().numpy().view(np.uint64).reshape(m, 4)
        return np.ascontiguousarray(np.concatenate([s, mm], axis=1).reshape(-1))

    H = [{k: host_aos(getattr(p, k)) for k in "xyabc"} for p in parties]
    keys = [p.key for p in parties]
    cores = host_cores()
    de = [ora.beaver_mask(FID, H[p]["x"], H[p]["y"], H[p]["a"], H[p]["b"]) for p in (0, 1)]   # the peers' d||e: untimed input
    res = [np.zeros(8 * m, dtype=np.uint64) for _ in (0, 1)]
    myde = [np.zeros(8 * m, dtype=np.uint64) for _ in (0, 1)]
    res_f = [np.zeros(8 * m, dtype=np.uint64) for _ in (0, 1)]
    myde_f = [np.zeros(8 * m, dtype=np.uint64) for _ in (0, 1)]
    P = ora._p

    def timed(fn, nthreads, o_de, o_res):
        t0 = time.perf_counter()
        for party in (0, 1):
            h = H[party]
            rc = fn(ctypes.c_int(FID), ctypes.c_size_t(m), ctypes.c_int(party), P(keys[party]), P(h["x"]), P(h["y"]), P(h["a"]), P(h["b"]),
                    P(h["c"]), P(de[1 - party]), P(o_de[party]), P(o_res[party]), ctypes.c_int(nthreads))
            assert rc == 0
        return time.perf_counter() - t0

    def mean_time(fn, nthreads, o_de, o_res, budget_s, max_reps):
        timed(fn, nthreads, o_de, o_res)  # warm
        reps, tot = 0, 0.0
        while tot < budget_s and reps < max_reps:
            tot += timed(fn, nthreads, o_de, o_res); reps += 1
        return tot / reps, reps

    nine, fused = ora.lib.ora_batch_mul_9pass_mt, ora.lib.ora_batch_mul_fused_mt
    t_all, reps = mean_time(nine, cores, myde, res, 4.0, 50)
    t_one, reps1 = mean_time(nine, 1, myde, res, 3.0, 8)
    tf_all, repsf = mean_time(fused, cores, myde_f, res_f, 3.0, 50)
    tf_one, repsf1 = mean_time(fused, 1, myde_f, res_f, 3.0, 8)
    assert np.array_equal(myde[0], de[0]) and np.array_equal(myde[1], de[1])
    same = all(np.array_equal(myde_f[p], myde[p]) and np.array_equal(res_f[p], res[p]) for p in (0, 1))
    assert same, "the fused CPU form disagrees with the nine passes"
    host = host_description()
    sample = "first 2^%d gates of the timed workload, both parties, 9-pass batch_mul, %d pthreads, mean of %d runs" % (int(np.log2(m)), cores, reps)
    detail = dict({
        "value": m / t_all, "unit": "gates/s", "cores": cores, "kind": "port",
        "label": "CPU restatement of reference algorithm (not ark-mpc measured)",
        "sample": "first 2^%d gates of the same seeded workload, both parties, the reference's literal 9-pass batch_mul "
                  "(oracle/ark_oracle.c ora_batch_mul_9pass_mt), %d pthreads static range split, mean of %d runs; "
                  "single_thread_value = 1 thread, mean of %d runs; fused_single_pass = the single-gate Mul's closure per element in one sweep "
                  "(ora_batch_mul_fused_mt, authenticated_scalar.rs:799-843), %d / %d runs" % (int(np.log2(m)), cores, reps, reps1, repsf, repsf1),
        "single_thread_value": m / t_one,
        "fused_single_pass": {"value": m / tf_all, "single_thread_value": m / tf_one, "unit": "gates/s", "cores": cores,
                              "same_words_as_nine_passes": bool(same)},
        "excludes": "the reference's DAG-executor overhead (13n+2 result slots per batch_mul, per-argument ResultValue clones, single_threaded.rs:322-356): "
                    "an optimistic stand-in for the reference, i.e. a conservative speed-up denominator (BASELINE.md section 3 step 3)",
    }, **host)
    # the headline line's object: numbers and short identifiers only (the prose is in the detail file)
    compact = {"value": m / t_all, "unit": "gates/s", "cores": cores, "kind": "port", "label": detail["label"], "sample": sample,
               "single_thread_value": m / t_one, "fused_value": m / tf_all, "fused_single_thread_value": m / tf_one,
               "cpu_model": host["cpu_model"], "nproc": host["nproc"], "compiler": host["compiler"], "flags": host["flags"],
               "cargo_probe": "absent" if host["cargo_probe"].startswith("absent") else host["cargo_probe"][:60]}
    return compact, detail, res, myde, m
