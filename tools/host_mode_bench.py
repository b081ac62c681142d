#!/usr/bin/env python3
"""PCIe-inclusive rate of the drop-in host-pointer mode (arkmpc_ctx_set_host_buffers): one party's K1 + K2/K3 on 2^20
gates with every buffer in pageable host memory (what a Rust Vec is).  Never the reported `value`; DESIGN.md section 6."""
import importlib, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
pkg = importlib.import_module("ark-mpc_amd")
n = 1 << int(os.environ.get("LOG2N", "20"))
e = pkg.Engine(0, device=0, host_buffers=True)
rng = np.random.default_rng(1)
def shares(): return (rng.integers(0, 2**63, size=8 * n, dtype=np.int64).view(np.uint64) >> np.uint64(3))   # < 2^253: valid residues
x, y, a, b, c = (shares() for _ in range(5))
peer_de = rng.integers(0, 2**63, size=8 * n, dtype=np.int64).view(np.uint64) >> np.uint64(3)
de = np.zeros(8 * n, dtype=np.uint64); out = np.zeros(8 * n, dtype=np.uint64)
key = np.array([5, 0, 0, 0], dtype=np.uint64)
def one():
    e.beaver_mask(n, x, y, a, b, de)
    e.beaver_finish_fused(n, 0, key, de, peer_de, a, b, c, out)
moved = n * (4 * 64 + 64 + 2 * 64 + 3 * 64 + 64)     # H2D x,y,a,b + D2H d||e + H2D d||e x2 + a,b,c + D2H out
def timed(label):
    one()
    t0 = time.perf_counter(); reps = 5
    for _ in range(reps): one()
    t = (time.perf_counter() - t0) / reps
    print(json.dumps({"mode": "host buffers (%s), one party, 2^%d gates, arkmpc_beaver_mask + arkmpc_beaver_finish_fused" % (label, int(np.log2(n))), "ms": t * 1e3,
                      "party_gates_per_s": n / t, "pcie_GBps": moved / t / 1e9}))
timed("pageable: whole-batch staging")
import ctypes
for arr in (x, y, a, b, c, peer_de, de, out):
    pkg.load_library().arkmpc_host_register(ctypes.c_void_p(arr.ctypes.data), ctypes.c_size_t(arr.nbytes))
timed("registered by the caller: three-stream pipeline")
for arr in (x, y, a, b, c, peer_de, de, out):
    pkg.load_library().arkmpc_host_unregister(ctypes.c_void_p(arr.ctypes.data))
# what the box's PCIe link gives with plain copies of the same volume (the ceiling for the mode above): pageable and pinned, one direction
# and both at once
import torch
m = 256 << 20
dev = torch.empty(m, dtype=torch.uint8, device="cuda"); dev2 = torch.empty(m, dtype=torch.uint8, device="cuda")
rows = {}
for kind in ("pageable", "pinned"):
    h = torch.empty(m, dtype=torch.uint8); h2 = torch.empty(m, dtype=torch.uint8)
    if kind == "pinned": h = h.pin_memory(); h2 = h2.pin_memory()
    h.fill_(1); h2.fill_(2)
    def h2d(): dev.copy_(h, non_blocking=True)
    def d2h(): h2.copy_(dev2, non_blocking=True)
    s2 = torch.cuda.Stream()
    def both():
        dev.copy_(h, non_blocking=True)
        with torch.cuda.stream(s2): h2.copy_(dev2, non_blocking=True)
    for name, fn in (("h2d", h2d), ("d2h", d2h), ("both_directions", both)):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(4): fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 4
        rows["%s_%s_GBps" % (kind, name)] = (m if name != "both_directions" else 2 * m) / dt / 1e9
print(json.dumps({"pcie_calibration_256MiB": rows}))
