"""What do arkmpc_host_register / _unregister cycles on ONE vector cost, and what does the runtime say about the vector in between?
(round 6: tests/pinning_scenarios.py measured 2 us per 64 MiB registration after the first cycle)"""
import ctypes, importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from helpers import FreshVA
pkg = importlib.import_module("ark-mpc_amd")
lib = pkg.load_library()
hip = ctypes.CDLL("libamdhip64.so")
eng = pkg.Engine(0, device=0)

class Attr(ctypes.Structure):
    _fields_ = [("type", ctypes.c_int), ("device", ctypes.c_int), ("devicePointer", ctypes.c_void_p), ("hostPointer", ctypes.c_void_p), ("isManaged", ctypes.c_int), ("flags", ctypes.c_uint)]

def knows(a):
    at = Attr()
    e = hip.hipPointerGetAttributes(ctypes.byref(at), ctypes.c_void_p(a.ctypes.data))
    return (e, at.type, hex(at.devicePointer or 0))

for mib in (4, 64):
    for how in ("arkmpc", "hip"):
        a = FreshVA.zeros((mib << 20) // 8); a.fill(1)
        P, B = ctypes.c_void_p(a.ctypes.data), ctypes.c_size_t(a.nbytes)
        print(mib, how, "before", knows(a))
        for cyc in range(4):
            t0 = time.perf_counter()
            rc1 = lib.arkmpc_host_register(P, B) if how == "arkmpc" else hip.hipHostRegister(P, B, 0)
            t1 = time.perf_counter()
            k1 = knows(a)
            t2 = time.perf_counter()
            rc2 = lib.arkmpc_host_unregister(P) if how == "arkmpc" else hip.hipHostUnregister(P)
            t3 = time.perf_counter()
            print(mib, how, cyc, "reg rc", rc1, "%.1f us" % ((t1 - t0) * 1e6), k1, "unreg rc", rc2, "%.1f us" % ((t3 - t2) * 1e6), "after", knows(a))
