"""integration/arkmpc_sys.rs (the Rust extern "C" surface for the ark-mpc shim of INTEGRATION.md) is generated from
include/arkmpc.h: it must be current and must declare every entry point the library exports, with the same arity."""
import importlib
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_generated_file_is_current():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_rust_ffi.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr + r.stdout


def test_every_entry_point_is_declared_with_matching_arity():
    eng = importlib.import_module("ark-mpc_amd.engine")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    gen = importlib.import_module("gen_rust_ffi")
    protos = {name: params for _, name, params in gen.parse(open(gen.HEADER).read())}
    assert sorted(protos) == eng.declared_symbols()
    rs = open(os.path.join(ROOT, "integration", "arkmpc_sys.rs")).read()
    for name, params in protos.items():
        m = re.search(r"pub fn %s\((.*?)\) -> " % name, rs)
        assert m, name
        got = [a for a in m.group(1).split(",") if a.strip()]
        assert len(got) == len(params), name
    for const in ("ARKMPC_OK: i32 = 0", "ARKMPC_ERR_NO_DEVICE: i32 = -4", "ARKMPC_BN254_FR: i32 = 0", "ARKMPC_WIRE_POINT_BATCH: i32 = 1"):
        assert const in rs


def test_rust_declarations_type_check_against_the_header(tmp_path):
    """integration/arkmpc_sys_check.c restates every Rust declaration with the ABI-equivalent C types and initialises a typed
    function pointer from the header's prototype: gcc rejects any mismatch in arity, integer width or pointer mutability."""
    src = os.path.join(ROOT, "integration", "arkmpc_sys_check.c")
    cmd = ["gcc", "-std=c11", "-fsyntax-only", "-Wall", "-Werror", "-Werror=incompatible-pointer-types", "-I", os.path.join(ROOT, "include"), src]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    # and it does catch drift: flip one mutability and one width
    txt = open(src).read()
    for good, bad in (("(arkmpc_ctx*, uintptr_t, const uint64_t*, const uint64_t*, uint64_t*) = arkmpc_scalar_add;",
                       "(arkmpc_ctx*, uintptr_t, uint64_t*, const uint64_t*, uint64_t*) = arkmpc_scalar_add;"),
                      ("static int32_t (*const chk_arkmpc_sync)(arkmpc_ctx*) = arkmpc_sync;", "static int64_t (*const chk_arkmpc_sync)(arkmpc_ctx*) = arkmpc_sync;")):
        assert good in txt, good
        mutated = tmp_path / "mutated.c"
        mutated.write_text(txt.replace(good, bad))
        r = subprocess.run(cmd[:-1] + [str(mutated)], capture_output=True, text=True)
        assert r.returncode != 0
