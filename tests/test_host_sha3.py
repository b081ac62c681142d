"""Host half of the hash commitment (H1): arkmpc_sha3_256 -- the sponge behind arkmpc_commit_sha3 -- against hashlib
(FIPS 202) for every inner loop the library carries (portable, 64-bit, 64-bit + BMI, AVX-512 planes / lanes / rows) and the automatic choice.
No GPU involved: the sponge is sequential by the reference's definition (commitment.rs:30-43) and runs on the host."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import ctypes, hashlib, importlib, sys
sys.path.insert(0, %r)
import numpy as np
lib = importlib.import_module("ark-mpc_amd").load_library()
lib.arkmpc_sha3_loop.restype = ctypes.c_char_p
rng = np.random.default_rng(11)
out = (ctypes.c_ubyte * 32)()
lens = list(range(0, 420)) + [543, 544, 545, 1087, 1088, 1089, 65536, 136 * 1000, 136 * 1000 + 135, (1 << 20) + 3]
for ln in lens:
    m = rng.integers(0, 256, ln, dtype=np.uint8).tobytes()
    assert lib.arkmpc_sha3_256(m, ctypes.c_size_t(ln), out) == 0
    assert bytes(out) == hashlib.sha3_256(m).digest(), ln
print("ok", len(lens), lib.arkmpc_sha3_loop().decode())
""" % ROOT


def _cpu_flags():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("flags"):
                return set(ln.split(":", 1)[1].split())
    except OSError:
        pass
    return set()


NEEDS = {"bmi": {"bmi1", "bmi2"}, "avx512": {"avx512f", "avx512vl"}, "lanes": {"avx512f", "avx512vl"}, "rows": {"avx512f", "avx512vl"}}


@pytest.mark.parametrize("loop", ["auto", "portable", "scalar", "bmi", "avx512", "lanes", "rows"])
def test_sha3_256_matches_hashlib(loop):
    """every loop is compared with hashlib BY NAME: the child reports which loop actually ran (arkmpc_sha3_loop), so a forced name that this
    CPU cannot run is a skip, not a silent second run of the timed choice"""
    env = dict(os.environ)
    env.pop("ARKMPC_KECCAK", None)
    if loop != "auto":
        env["ARKMPC_KECCAK"] = loop
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert r.stdout.startswith("ok")
    ran = r.stdout.split()[2]
    if loop == "auto":
        assert ran in ("portable", "scalar", "bmi", "avx512", "lanes", "rows")
    elif ran != loop:
        missing = NEEDS.get(loop, set()) - _cpu_flags()
        assert missing, "ARKMPC_KECCAK=%s ran %r although this CPU has %s" % (loop, ran, sorted(NEEDS.get(loop, set())))
        pytest.skip("this CPU lacks %s: the %s loop cannot run here (the timed choice %r ran instead)" % (sorted(missing), loop, ran))
