#!/usr/bin/env python3
"""hipHostRegister / copy / hipHostUnregister churn over 16 x 64 MiB numpy buffers, eight per round: what pinning a session's buffers in place costs when
sessions follow each other (register 0.28 ms per 64 MiB, unregister < 0.01 ms; buffers seen before re-register in microseconds)."""
import ctypes, importlib, sys, time, os
import numpy as np
sys.path.insert(0, "/root/repo")
pkg = importlib.import_module("ark-mpc_amd"); lib = pkg.load_library()
e = pkg.Engine(0, device=0)
import torch
bufs = [np.ones(8 << 20, dtype=np.uint64) for _ in range(16)]
dev = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
P = lambda a: ctypes.c_void_p(a.ctypes.data)
for rnd in range(4):
    tr = tu = tc = 0.0
    for k in range(8):
        a = bufs[(8 * rnd + k) % 16]
        t0 = time.perf_counter(); lib.arkmpc_host_register(P(a), ctypes.c_size_t(a.nbytes)); t1 = time.perf_counter()
        e.call("memcpy_h2d", dev, a, ("size", a.nbytes)); t2 = time.perf_counter()
        tr += t1 - t0; tc += t2 - t1
    for k in range(8):
        a = bufs[(8 * rnd + k) % 16]
        t0 = time.perf_counter(); lib.arkmpc_host_unregister(P(a)); tu += time.perf_counter() - t0
    print("round %d: register %.3f ms per 64 MiB, copy %.3f ms, unregister %.3f ms" % (rnd, tr / 8 * 1e3, tc / 8 * 1e3, tu / 8 * 1e3))
