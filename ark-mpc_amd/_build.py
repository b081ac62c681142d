"""Build recipe for the engine's native libraries (hipcc, gfx950 only; no GPU needed to build).

    python ark-mpc_amd/_build.py [--force]      (also driven by __graft_entry__.build())

Outputs (git-ignored, but they travel to the GPU box with the tree):
    ark-mpc_amd/lib/libarkmpc_hip.so    the C-ABI engine (include/arkmpc.h)
    ark-mpc_amd/lib/libarkmpc_host.so   the C++ host-side mirror of the reference's fabric API
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
HOST = os.path.join(HERE, "host")
LIB = os.path.join(HERE, "lib")
OBJ = os.path.join(HERE, "lib", "obj")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"
HIP_FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]

HOST_CPP_SOURCES = ["keccak_avx512.cpp"]
ENGINE_SOURCES = ["arkmpc_scalar.hip", "arkmpc_group.hip", "arkmpc_curve.hip", "arkmpc_edwards.hip", "arkmpc_wire.hip", "arkmpc_batch.hip", "sha3_host.hip"]
# every header / include fragment in csrc (a stale .so after editing an .inc is the failure this guards against)
ENGINE_DEPS = sorted(f for f in os.listdir(CSRC) if f.endswith((".inc", ".hpp", ".json"))) + [os.path.join("..", "..", "include", "arkmpc.h")]


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def _run(cmd):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("build failed: %s\n%s" % (" ".join(cmd), r.stdout))
    return r.stdout


def build_engine(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    deps = [os.path.join(CSRC, d) for d in ENGINE_DEPS]
    objs, jobs = [], []
    for src in ENGINE_SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _newer(o, [s] + deps):
            jobs.append([HIPCC] + HIP_FLAGS + ["-c", s, "-o", o])
    for src in HOST_CPP_SOURCES:          # plain host C++ (per-function target attributes; no device pass)
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src.replace(".cpp", ".o"))
        objs.append(o)
        if force or _newer(o, [s]):
            # ROCm's clang++ schedules the AVX-512 Keccak loops better than g++ 11 (0.87 vs 0.82 GB/s on EPYC 9575F); g++ if it is not there
            cxx = os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib", "llvm", "bin", "clang++")
            jobs.append([cxx if os.path.exists(cxx) else "g++", "-O3", "-std=c++17", "-fPIC", "-Wall", "-c", s, "-o", o])
    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            for out in ex.map(_run, jobs):
                if verbose and out.strip():
                    print(out)
    so = os.path.join(LIB, "libarkmpc_hip.so")
    if force or jobs or _newer(so, objs):
        _run([HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", so] + objs)
    # test-only entry points (include/arkmpc_test_hooks.h) that must not ship in the product library: their own small shared object
    hooks_src, hooks_so = os.path.join(CSRC, "arkmpc_testhooks.hip"), os.path.join(LIB, "libarkmpc_testhooks.so")
    if force or _newer(hooks_so, [hooks_src] + deps):
        _run([HIPCC] + HIP_FLAGS + ["-shared", "-o", hooks_so, hooks_src])
    return so


def build_hazard(force=False, verbose=False):
    """lib/libarkmpc_hip_hazard.so: the same engine compiled with -DARKMPC_HAZARD_SWITCHES, which brings back the two environment switches of the
    round-5 hazard hunt -- ARKMPC_ZC_ON_OWN_PINS=1 (kernels address vectors the library registered itself: the known-bad combination) and
    ARKMPC_PIN_DRAIN -- that the product library does not contain.  Built on demand only (tools/crash_hunt.sh, probes/); ARKMPC_LIBRARY selects it."""
    obj_dir = os.path.join(LIB, "obj_hazard")
    os.makedirs(obj_dir, exist_ok=True)
    deps = [os.path.join(CSRC, d) for d in ENGINE_DEPS]
    objs, jobs = [], []
    for src in ENGINE_SOURCES:
        s, o = os.path.join(CSRC, src), os.path.join(obj_dir, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _newer(o, [s] + deps):
            jobs.append([HIPCC] + HIP_FLAGS + ["-DARKMPC_HAZARD_SWITCHES", "-c", s, "-o", o])
    objs += [os.path.join(OBJ, src.replace(".cpp", ".o")) for src in HOST_CPP_SOURCES]       # (host objects: shared with the product build)
    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            for out in ex.map(_run, jobs):
                if verbose and out.strip():
                    print(out)
    so = os.path.join(LIB, "libarkmpc_hip_hazard.so")
    if force or jobs or _newer(so, objs):
        _run([HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", so] + objs)
    return so


def build_host(force=False, verbose=False):
    """C++ host-side mirror of the reference's fabric API (host/fabric.hpp, header-only) + its two-party driver."""
    src = os.path.join(HOST, "mock_mpc_main.cpp")
    if not os.path.exists(src):
        return None
    exe = os.path.join(LIB, "arkmpc_mock_mpc")
    deps = [src, os.path.join(HOST, "fabric.hpp"), os.path.join(HERE, "..", "include", "arkmpc.h"), os.path.join(LIB, "libarkmpc_hip.so")]
    for s_, e_ in ((src, exe), (os.path.join(HOST, "bench_main.cpp"), os.path.join(LIB, "arkmpc_host_bench"))):
        if os.path.exists(s_) and (force or _newer(e_, [s_] + deps[1:])):
            _run(["g++", "-O2", "-std=c++17", "-Wall", "-pthread", "-I", os.path.join(HERE, "..", "include"), "-I", HOST, "-o", e_, s_,
                  "-L", LIB, "-larkmpc_hip", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath," + os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib")])
    return exe


def build_all(force=False, verbose=False):
    a = build_engine(force, verbose)
    b = build_host(force, verbose)
    return a, b


if __name__ == "__main__":
    print(build_all(force="--force" in sys.argv, verbose=True))
    if "--hazard" in sys.argv:
        print(build_hazard(force="--force" in sys.argv, verbose=True))
