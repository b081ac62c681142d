#!/usr/bin/env python3
"""What arkmpc_host_alloc / arkmpc_host_free (hipHostMalloc / hipHostFree) cost per call, by size -- for a shim that would allocate its operand
vectors pinned.   python probes/host_alloc_cost_probe.py"""
import ctypes, importlib, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("ark-mpc_amd")
lib = pkg.load_library()
eng = pkg.Engine(0, device=0)
for mib in (1, 16, 64, 256):
    ta, tf = [], []
    for _ in range(6):
        q = ctypes.c_void_p()
        t0 = time.perf_counter(); rc = lib.arkmpc_host_alloc(ctypes.c_size_t(mib << 20), ctypes.byref(q)); t1 = time.perf_counter()
        assert rc == 0
        ctypes.memset(q, 1, mib << 20)
        t2 = time.perf_counter(); lib.arkmpc_host_free(q); t3 = time.perf_counter()
        ta.append(t1 - t0); tf.append(t3 - t2)
    ta.sort(); tf.sort()
    print(json.dumps({"MiB": mib, "host_alloc_ms": round(ta[len(ta) // 2] * 1e3, 3), "host_free_ms": round(tf[len(tf) // 2] * 1e3, 3)}))
