#!/usr/bin/env python3
"""Does the socket a host buffer lives on change the DMA rate to the GPU?  Allocates + first-touches 64 MiB numpy buffers with the thread bound to
the CPUs of each NUMA node in turn, pins them (arkmpc_host_register) and times plain H2D / D2H copies through the C ABI."""
import ctypes, glob, importlib, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
pkg = importlib.import_module("ark-mpc_amd")
lib = pkg.load_library()
e = pkg.Engine(0, device=0)
import torch
def cpus_of(node):
    s = open("/sys/devices/system/node/node%d/cpulist" % node).read().strip()
    out = []
    for part in s.split(","):
        a, _, b = part.partition("-")
        out += list(range(int(a), int(b or a) + 1))
    return out
nodes = sorted(int(p.rsplit("node", 1)[1]) for p in glob.glob("/sys/devices/system/node/node[0-9]*"))
gpu_nodes = {}
for p in glob.glob("/sys/class/drm/card*/device/numa_node"):
    gpu_nodes[p.split("/")[4]] = open(p).read().strip()
print(json.dumps({"numa_nodes": nodes, "gpu_numa_node": gpu_nodes, "affinity_at_start": len(os.sched_getaffinity(0))}))
nb = 64 << 20
dev = torch.empty(nb, dtype=torch.uint8, device="cuda")
all_cpus = os.sched_getaffinity(0)
for node in nodes:
    os.sched_setaffinity(0, set(cpus_of(node)) & all_cpus or all_cpus)
    bufs = []
    for k in range(3):
        a = np.empty(nb // 8, dtype=np.uint64); a.fill(k + 1)
        lib.arkmpc_host_register(ctypes.c_void_p(a.ctypes.data), ctypes.c_size_t(a.nbytes))
        bufs.append(a)
    rates = []
    for a in bufs:
        for direction in ("h2d", "d2h"):
            fn = (lambda: e.call("memcpy_h2d", dev, a, ("size", nb))) if direction == "h2d" else (lambda: e.call("memcpy_d2h", a, dev, ("size", nb)))
            fn()
            t0 = time.perf_counter()
            for _ in range(4): fn()
            rates.append((direction, nb * 4 / (time.perf_counter() - t0) / 1e9))
    print(json.dumps({"buffers_first_touched_on_node": node, "GBps": [(d, round(r, 1)) for d, r in rates]}))
    for a in bufs:
        lib.arkmpc_host_unregister(ctypes.c_void_p(a.ctypes.data))
os.sched_setaffinity(0, all_cpus)
