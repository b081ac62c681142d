#!/usr/bin/env python3
"""Two host threads with a context each run INDEPENDENT one-party sessions back to back on caller-pinned vectors (a rayon executor evaluating two
batch_mul gates at once): does the aggregate beat one worker's 1.30e8 party-gates/s?"""
import ctypes, importlib, json, os, sys, threading, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from benchlib import common as bench
pkg = importlib.import_module("ark-mpc_amd"); lib = pkg.load_library()
torch.cuda.set_device(0)
eng = pkg.Engine(0, device=0, stream=torch.cuda.current_stream().cuda_stream)
n = 1 << 20
parties, _ = bench.build_workload(eng, n, seed=5, layout="aos")
bench.step(bench.prepare_step(eng, n, parties, "aos")); torch.cuda.synchronize()
host = lambda t: np.ascontiguousarray(t.cpu().numpy().view(np.uint64))
H = [{k: host(getattr(p, k)) for k in "xyabc"} for p in parties]
de = [host(p.de) for p in parties]; want = [host(p.out) for p in parties]; keys = [p.key for p in parties]
W = 2
es = [pkg.Engine(0, device=0) for _ in range(W)]
bufs = [(np.zeros(8 * n, dtype=np.uint64), np.zeros(8 * n, dtype=np.uint64)) for _ in range(W)]
for a in [v for h in H for v in h.values()] + de + [b for t in bufs for b in t]:
    lib.arkmpc_host_register(ctypes.c_void_p(a.ctypes.data), ctypes.c_size_t(a.nbytes))
reps = 8
bar = threading.Barrier(W)
def worker(w):
    torch.cuda.set_device(0)
    p = w & 1
    for r in range(reps + 1):
        if r == 1: bar.wait()
        s = es[w].hostmul_begin(n, H[p]["x"], H[p]["y"], H[p]["a"], H[p]["b"], H[p]["c"], bufs[w][0])
        es[w].hostmul_finish(s, p, keys[p], de[1 - p], bufs[w][1])
for workers in (1, 2):
    W_run = workers
    bar = threading.Barrier(W_run)
    th = [threading.Thread(target=worker, args=(w,)) for w in range(W_run)]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    dt = time.perf_counter() - t0
    ok = all(np.array_equal(bufs[w][1], want[w & 1]) for w in range(W_run))
    print(json.dumps({"workers": W_run, "sessions_each": reps + 1, "aggregate_party_gates_per_s": W_run * (reps + 1) * n / dt, "ms_per_session_per_worker": dt / (reps + 1) * 1e3, "ok": ok}))
