#!/usr/bin/env python3
"""Where a wire-form session's time goes: host-side spans of arkmpc_hostmul_begin_wire / _finish_wire at 2^20 gates on pinned buffers
(run under `rocprofv3 --kernel-trace --memory-copy-trace` for the device side).   python probes/wire_session_probe.py [log2n]"""
import ctypes, importlib, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("ark-mpc_amd")
lib = pkg.load_library()
L = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << L
eng = pkg.Engine(0, device=0)
rng = np.random.default_rng(1)

def pinned(nbytes, dtype):
    q = ctypes.c_void_p()
    assert lib.arkmpc_host_alloc(ctypes.c_size_t(nbytes), ctypes.byref(q)) == 0
    ct = ctypes.c_uint64 if dtype == np.uint64 else ctypes.c_uint8
    return np.ctypeslib.as_array(ctypes.cast(q, ctypes.POINTER(ct)), shape=(nbytes // np.dtype(dtype).itemsize,))

def shares():
    a = pinned(64 * n, np.uint64)
    a[:] = rng.integers(0, 2**63, size=8 * n, dtype=np.int64).view(np.uint64) >> np.uint64(3)     # < 2^253: valid residues
    return a
x, y, a, b, c = (shares() for _ in range(5))
out = pinned(64 * n, np.uint64)
cap = eng.wire_frame_bound(2 * n)
frame, peer = pinned(cap, np.uint8), pinned(cap, np.uint8)
key = np.array([5, 0, 0, 0], dtype=np.uint64)
s, ln = eng.hostmul_begin_wire(n, x, y, a, b, c, 1, peer); eng.hostmul_abort(s)      # a valid inbound frame: our own
rows = []
for k in range(6):
    t0 = time.perf_counter()
    s, l2 = eng.hostmul_begin_wire(n, x, y, a, b, c, 1, frame)
    t1 = time.perf_counter()
    eng.hostmul_finish_wire(s, 0, key, peer, ln, out)
    t2 = time.perf_counter()
    rows.append([round((t1 - t0) * 1e3, 3), round((t2 - t1) * 1e3, 3)])
print(json.dumps({"log2n": L, "frame_bytes": ln, "begin_wire_ms, finish_wire_ms": rows}))
# two workers (a context and a thread each, own buffers): does one's frame download hide under the other's uploads?
import threading
def mk():
    v = [shares() for _ in range(5)]
    return v, pinned(64 * n, np.uint64), pinned(cap, np.uint8), pkg.Engine(0, device=0)
workers = [mk() for _ in range(2)]
reps = 8
def work(w):
    v, o, fr, e = workers[w]
    for _ in range(reps):
        s_, _l = e.hostmul_begin_wire(n, v[0], v[1], v[2], v[3], v[4], 1, fr)
        e.hostmul_finish_wire(s_, 0, key, peer, ln, o)
for w in (0, 1):
    work(w)
t0 = time.perf_counter()
th = [threading.Thread(target=work, args=(w,)) for w in (0, 1)]
for t in th: t.start()
for t in th: t.join()
t = time.perf_counter() - t0
print(json.dumps({"two_workers": {"sessions": 2 * reps, "ms_per_session_aggregate": t / (2 * reps) * 1e3, "party_gates_per_s": 2 * reps * n / t}}))
