#!/usr/bin/env python3
"""Host SHA3-256 sponge rate (the H1 floor of config 5) for each inner loop of csrc/keccak_avx512.cpp and the automatic choice."""
import ctypes, hashlib, importlib, json, os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "child":
    pkg = importlib.import_module("ark-mpc_amd")
    lib = pkg.load_library()
    m = np.random.default_rng(1).integers(0, 256, 1 << 28, dtype=np.uint8).tobytes()
    out = (ctypes.c_ubyte * 32)()
    lib.arkmpc_sha3_256(m, ctypes.c_size_t(1 << 20), out)
    t0 = time.perf_counter(); lib.arkmpc_sha3_256(m, ctypes.c_size_t(len(m)), out); t = time.perf_counter() - t0
    t1 = time.perf_counter(); want = hashlib.sha3_256(m).digest(); th = time.perf_counter() - t1
    assert bytes(out) == want
    print(json.dumps({"inner_loop": os.environ.get("ARKMPC_KECCAK", "auto (timed at first use)"), "GBps": len(m) / t / 1e9,
                      "hashlib_openssl_GBps": len(m) / th / 1e9}))
else:
    for env in ({}, {"ARKMPC_KECCAK": "portable"}, {"ARKMPC_KECCAK": "scalar"}, {"ARKMPC_KECCAK": "bmi"}, {"ARKMPC_KECCAK": "avx512"}, {"ARKMPC_KECCAK": "lanes"}, {"ARKMPC_KECCAK": "rows"}):
        print(subprocess.run([sys.executable, __file__, "child"], env={**os.environ, **env}, capture_output=True, text=True).stdout.strip())
