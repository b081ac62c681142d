#!/usr/bin/env python3
"""What a pinned block costs from the runtime (hipHostMalloc / hipHostFree, reached through an empty free list / arkmpc_host_trim) and from the
free list of arkmpc_host_alloc / arkmpc_host_free, by size -- for a shim that would allocate its operand
vectors pinned.   python probes/host_alloc_cost_probe.py"""
import ctypes, importlib, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("ark-mpc_amd")
lib = pkg.load_library()
eng = pkg.Engine(0, device=0)
lib.arkmpc_host_trim()
for mib in (1, 16, 64, 256):
    ta, tf = [], []
    for _ in range(6):
        lib.arkmpc_host_trim()                                 # (every round a fresh block from the runtime: the list below the loop is the recycled case)
        q = ctypes.c_void_p()
        t0 = time.perf_counter(); rc = lib.arkmpc_host_alloc(ctypes.c_size_t(mib << 20), ctypes.byref(q)); t1 = time.perf_counter()
        assert rc == 0
        ctypes.memset(q, 1, mib << 20)
        lib.arkmpc_host_free(q)                                # onto the free list ...
        t2 = time.perf_counter(); lib.arkmpc_host_trim(); t3 = time.perf_counter()      # ... and back to the runtime: hipHostFree
        ta.append(t1 - t0); tf.append(t3 - t2)
    ta.sort(); tf.sort()
    lib.arkmpc_host_trim()
    q = ctypes.c_void_p(); lib.arkmpc_host_alloc(ctypes.c_size_t(mib << 20), ctypes.byref(q)); lib.arkmpc_host_free(q)      # one block of this class on the free list
    tr = []
    for _ in range(50):
        t0 = time.perf_counter(); lib.arkmpc_host_alloc(ctypes.c_size_t(mib << 20), ctypes.byref(q)); lib.arkmpc_host_free(q); tr.append(time.perf_counter() - t0)
    tr.sort()
    print(json.dumps({"MiB": mib, "runtime_alloc_ms": round(ta[len(ta) // 2] * 1e3, 3), "runtime_free_ms": round(tf[len(tf) // 2] * 1e3, 3),
                      "recycled_alloc_plus_free_us": round(tr[len(tr) // 2] * 1e6, 2)}))
