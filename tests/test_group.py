"""The multi-device group of the C ABI (include/arkmpc.h arkmpc_group_*, csrc/arkmpc_group.hip): ONE process, G members.

On the one-GPU test box the members share device 0 (repeated ids), so the sharded path -- range kernels, peer pushes, gathers, the
pipelined commitment, the AND-reduced verify flag -- runs for real and must equal the unsharded single-context result (C99 caller
tests/c/group_oversub.c) and the oracle (the Python binding below).  Without a GPU the C caller must fail loudly."""
import subprocess

import numpy as np
import pytest

import pyref
from helpers import authenticated_shares, mont_array, rand_values
from test_abi_cpu import _build_c_smoke


def test_group_c_caller_builds_and_fails_loudly_without_gpu(tmp_path):
    import torch
    exe = _build_c_smoke(tmp_path, "group_oversub")
    r = subprocess.run([exe, "1000", "4"], capture_output=True, text=True)
    if torch.cuda.is_available():
        assert r.returncode == 0 and "group ok" in r.stdout, r.stdout + r.stderr
    else:
        assert r.returncode == 3 and "no device" in r.stdout


def test_group_create_without_gpu_is_no_device(pkg):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(pkg.ArkMpcError):
        pkg.Group("bn254_fr", [0, 0])


def test_shard_ranges_match_sharding_py():
    """arkmpc_group_shard_range is specified as sharding.shard_range; the formula is restated here for sizes the GPU test then checks."""
    import importlib
    sh = importlib.import_module("ark-mpc_amd.sharding")
    for n in (0, 1, 5, 1000, 100003, 1 << 24):
        for world in (1, 2, 3, 8):
            got = [sh.shard_range(n, world, r) for r in range(world)]
            assert got[0][0] == 0 and got[-1][1] == n and all(got[i][1] == got[i + 1][0] for i in range(world - 1))


@pytest.mark.gpu
@pytest.mark.parametrize("n,G", [(100003, 4), (1000, 3), (5, 8), (1, 2), (0, 2), (257, 1)])
def test_group_oversubscribed_equals_single_context(tmp_path, n, G):
    """configs 3 and 5 in shape (ragged sizes, more members than elements, empty batch): every buffer bit-equal to ONE context."""
    r = subprocess.run([_build_c_smoke(tmp_path, "group_oversub"), str(n), str(G)], capture_output=True, text=True)
    assert r.returncode == 0 and "group ok" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_group_peer_copy_api_on_one_device(tmp_path):
    """ARKMPC_GROUP_FORCE_PEER=1: the gathers go through hipMemcpyPeerAsync (the call the multi-GPU path makes) even though the members share
    device 0 -- same results."""
    import os
    env = dict(os.environ, ARKMPC_GROUP_FORCE_PEER="1")
    r = subprocess.run([_build_c_smoke(tmp_path, "group_oversub"), "20011", "4"], capture_output=True, text=True, env=env)
    assert r.returncode == 0 and "group ok" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_group_config3_and_config5_shapes_8_members(tmp_path):
    """8 members (the 8 GPUs of BASELINE configs 3 and 5, here sharing one), 2^21 gates / shares: bit-equal to ONE context."""
    r = subprocess.run([_build_c_smoke(tmp_path, "group_oversub"), str(1 << 21), "8"], capture_output=True, text=True)
    assert r.returncode == 0 and "group ok" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("layout", [0, 1])
def test_group_beaver_vs_oracle_through_python_binding(pkg, oracle, layout):
    """Two parties, each a 3-member group on device 0, Beaver batch_mul of 1000 gates: d||e and the product records vs the oracle."""
    fid, n, G = 0, 1000, 3
    p = pyref.P[fid]
    k0, k1 = rand_values(fid, 2, 1)
    key = (k0 + k1) % p
    keys = [mont_array(fid, [k0]), mont_array(fid, [k1])]
    x, y, a, b = (rand_values(fid, n, s) for s in (2, 3, 4, 5))
    c = [(u * v) % p for u, v in zip(a, b)]
    sh = {nm: authenticated_shares(fid, v, key, 10 + i) for i, (nm, v) in enumerate(zip("xyabc", (x, y, a, b, c)))}
    grp = [pkg.Group(fid, [0] * G) for _ in (0, 1)]
    segs, ew = (2, 4) if layout else (1, 8)
    S = [{nm: grp[pid].malloc(n, segs, ew) for nm in "xyabco"} for pid in (0, 1)]
    de = [grp[pid].malloc(n, 2, 4) for pid in (0, 1)]
    for pid in (0, 1):
        for nm in "xyabc":
            grp[pid].shares_from_host(layout, n, sh[nm][pid], S[pid][nm])
        grp[pid].beaver_mask(layout, n, S[pid]["x"], S[pid]["y"], S[pid]["a"], S[pid]["b"], de[pid])
    ode = [oracle.beaver_mask(fid, sh["x"][pid], sh["y"][pid], sh["a"][pid], sh["b"][pid]) for pid in (0, 1)]
    opened = oracle.open_combine(fid, ode[0], ode[1])
    for pid in (0, 1):
        got = np.zeros(8 * n, dtype=np.uint64)
        grp[pid].gather_d2h(n, 2, 4, de[pid], got)
        assert np.array_equal(got, ode[pid])
    for pid in (0, 1):
        grp[pid].beaver_finish_fused(layout, n, pid, keys[pid], de[pid], de[1 - pid], S[pid]["a"], S[pid]["b"], S[pid]["c"], S[pid]["o"])
        got = np.zeros(8 * n, dtype=np.uint64)
        grp[pid].shares_to_host(layout, n, S[pid]["o"], got)
        want = oracle.beaver_finish(fid, pid, keys[pid], opened[:4 * n].copy(), opened[4 * n:].copy(), sh["a"][pid], sh["b"][pid], sh["c"][pid])
        assert np.array_equal(got, want)
    for pid in (0, 1):
        lo_cnt = [grp[pid].shard_range(n, m) for m in range(G)]
        assert lo_cnt == [((n * m) // G, (n * (m + 1)) // G - (n * m) // G) for m in range(G)]
        for v in list(S[pid].values()) + [de[pid]]:
            grp[pid].free(v)
        grp[pid].close()


@pytest.mark.gpu
def test_group_open_authenticated_vs_oracle(pkg, oracle):
    """config 5 in shape through the binding: BLS12-381 shares over 4 members: opened values, MAC-check shares, commitment (vs the
    oracle's SHA3 of the ordered byte stream) and the verify flag with one corrupted MAC."""
    fid, n, G = 1, 777, 4
    p = pyref.P[fid]
    k0, k1 = rand_values(fid, 2, 21)
    key = (k0 + k1) % p
    keys = [mont_array(fid, [k0]), mont_array(fid, [k1])]
    v = rand_values(fid, n, 22)
    sh = authenticated_shares(fid, v, key, 23)
    blind = [mont_array(fid, [rand_values(fid, 1, 30 + pid)[0]]) for pid in (0, 1)]
    for corrupt in (False, True):
        shares = [s.copy() for s in sh]
        if corrupt:
            shares[1][8 * (n - 1) + 4] ^= np.uint64(1)                      # last MAC of party 1
        grp = [pkg.Group(fid, [0] * G) for _ in (0, 1)]
        S = [grp[pid].malloc(n, 1, 8) for pid in (0, 1)]
        mine = [grp[pid].malloc(n, 1, 4) for pid in (0, 1)]
        op = [grp[pid].malloc(n, 1, 4) for pid in (0, 1)]
        chk = [grp[pid].malloc(n, 1, 4) for pid in (0, 1)]
        for pid in (0, 1):
            grp[pid].shares_from_host(0, n, shares[pid], S[pid])
            grp[pid].share_extract(0, n, S[pid], mine[pid])
            grp[pid].sync()
        ext = lambda a: np.ascontiguousarray(a.reshape(-1, 8)[:, :4]).reshape(-1)         # the `.share()` projection
        want_open = oracle.open_combine(fid, ext(shares[0]), ext(shares[1]))
        for pid in (0, 1):
            grp[pid].open_and_mac_check(0, n, keys[pid], S[pid], mine[1 - pid], op[pid], chk[pid])
            got = np.zeros(4 * n, dtype=np.uint64)
            grp[pid].gather_d2h(n, 1, 4, op[pid], got)
            assert np.array_equal(got, want_open)
            want_chk = oracle.mac_check_shares(fid, keys[pid], want_open, shares[pid])
            grp[pid].gather_d2h(n, 1, 4, chk[pid], got)
            assert np.array_equal(got, want_chk)
            assert np.array_equal(grp[pid].commit_sha3(n, chk[pid], blind[pid]), oracle.commit_scalars(fid, want_chk, blind[pid]))
        for pid in (0, 1):
            grp[pid].sync()
        assert grp[0].mac_verify(n, chk[0], chk[1]) == (not corrupt)
        assert grp[0].mac_verify(n, chk[0], chk[1]) == (not corrupt)            # the flag is collected and cleared per call
        for pid in (0, 1):
            for vec in (S[pid], mine[pid], op[pid], chk[pid]):
                grp[pid].free(vec)
            grp[pid].close()
