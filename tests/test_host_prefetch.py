"""What the C++ host mirror (ark-mpc_amd/host/fabric.hpp) does with triples it has read AHEAD of need (MpcFabric::prefetch_triples), under fault
injection: tests/cpp/prefetch_fault.cpp interposes arkmpc_batch_from_host_async and fails one call of the caller's choosing."""
import os
import subprocess

import pytest


def _build_prefetch_fault(tmp_path):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe, lib = str(tmp_path / "prefetch_fault"), os.path.join(root, "ark-mpc_amd", "lib")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-Wall", "-pthread", "-I", os.path.join(root, "include"), "-I", os.path.join(root, "ark-mpc_amd", "host"),
                           "-o", exe, os.path.join(root, "tests", "cpp", "prefetch_fault.cpp"), "-L", lib, "-larkmpc_hip", "-ldl",
                           "-Wl,-rpath," + lib, "-Wl,-rpath," + os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib")])
    return exe


def test_prefetch_fault_program_builds(tmp_path):
    """tests/cpp/prefetch_fault.cpp (fault injection by symbol interposition over the header-only host mirror) compiles and links here;
    without a GPU it fails loudly"""
    import torch
    r = subprocess.run([_build_prefetch_fault(tmp_path), "100", "2", "300", "-1", "0"], capture_output=True, text=True, timeout=120)
    if not torch.cuda.is_available():
        assert r.returncode == 1 and "no CPU fallback" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("link,layout", [("host", "split"), ("device", "aos")])
def test_triples_read_ahead_survive_a_failed_upload_and_the_tail_is_not_burned(tmp_path, link, layout):
    """Round-5 advisor finding on MpcFabric::prefetch_triples: (i) triples the read-ahead CONSUMED from the source must not be dropped when their
    asynchronous import fails -- this party's FIFO would then run n ahead of its peer's and every later gate would fail its MAC check far from the
    cause; (ii) the read-ahead after the LAST gate burned n real triples for nothing.  Now: a failed import leaves the records with the fabric
    and they go up blocking at need; a source that says it cannot serve n more is not asked ahead; set_triple_prefetch(false) before the last
    gate reads nothing ahead."""
    exe = _build_prefetch_fault(tmp_path)
    n, gates = 20000, 3

    def run_(cap, fail, hint):
        r = subprocess.run([exe, str(n), str(gates), str(cap), str(fail), str(hint)], capture_output=True, text=True, timeout=120,
                           env=dict(os.environ, ARKMPC_MOCK_LINK=link, ARKMPC_SHARE_LAYOUT=layout))
        assert r.returncode == 0 and r.stdout.startswith("ok "), r.stdout + r.stderr
        _, c0, c1, chk, calls = r.stdout.split()
        return int(c0), int(c1), chk, int(calls)

    c0, c1, want, calls = run_(4 * n, -1, 0)
    assert (c0, c1) == (4 * n, 4 * n) and calls == 2 * 3 * 4          # no hint, a source with triples to spare: one batch is read ahead of a gate that never comes
    if link == "host":
        assert run_(4 * n, -1, 1)[:3] == (3 * n, 3 * n, want)         # the caller marked its last gate: nothing consumed beyond the circuit
        assert run_(3 * n, -1, 0)[:3] == (3 * n, 3 * n, want)         # a source with exactly enough: it is not asked ahead for what it cannot serve
    for fail in ((0, 7, calls - 1) if link == "host" else (13,)):     # an import fails (both parties count; any of a, b, c; first fetch, read-ahead, tail)
        got = run_(4 * n, fail, 0)
        assert got[:3] == (4 * n, 4 * n, want), (fail, got)
