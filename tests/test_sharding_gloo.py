"""N>1 path on CPU: world_size-2 gloo.  Each rank evaluates its contiguous gate range of a two-party Beaver
batch_mul + authenticated open (compute stand-in on CPU = the oracle, since the engine needs a GPU), then the
opened values / MAC-check shares are all-gathered in order and the verify flag is AND-reduced.  The sharded
result must equal the unsharded one bit for bit."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, n, corrupt, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_api
    from helpers import authenticated_shares, mont_array, rand_values
    import pyref
    sharding = importlib.import_module("ark-mpc_amd.sharding")
    ora = oracle_api.load()
    fid = 0
    p = pyref.P[fid]
    k0, k1 = rand_values(fid, 2, 1); key = (k0 + k1) % p
    keys = [mont_array(fid, [k0]), mont_array(fid, [k1])]
    x, y, a, b = (rand_values(fid, n, s) for s in (2, 3, 4, 5))
    c = [(u * v) % p for u, v in zip(a, b)]
    sh = {nm: authenticated_shares(fid, v, key, 10 + i) for i, (nm, v) in enumerate(zip("xyabc", (x, y, a, b, c)))}
    if corrupt:
        sh["c"][0][8 * (n - 1) + 4] ^= np.uint64(1)     # flips a MAC limb in the LAST shard only
    lo, hi = sharding.shard_range(n, world, rank)
    sl = lambda arr: np.ascontiguousarray(arr[8 * lo:8 * hi])
    m = hi - lo
    de = [ora.beaver_mask(fid, sl(sh["x"][q_]), sl(sh["y"][q_]), sl(sh["a"][q_]), sl(sh["b"][q_])) for q_ in (0, 1)]
    opened = ora.open_combine(fid, de[0], de[1])
    res = [ora.beaver_finish(fid, q_, keys[q_], opened[:4 * m].copy(), opened[4 * m:].copy(), sl(sh["a"][q_]), sl(sh["b"][q_]), sl(sh["c"][q_])) for q_ in (0, 1)]
    mine = [np.ascontiguousarray(r.reshape(-1, 8)[:, :4].reshape(-1)) for r in res]
    val = ora.open_combine(fid, mine[0], mine[1])
    chk = [ora.mac_check_shares(fid, keys[q_], val, res[q_]) for q_ in (0, 1)]
    ok_local = ora.mac_verify(fid, chk[0], chk[1])
    t = lambda arr: torch.from_numpy(arr.view(np.int64).copy())
    full_val = sharding.gather_ordered(t(val), n, 4).numpy().view(np.uint64)
    full_chk0 = sharding.gather_ordered(t(chk[0]), n, 4).numpy().view(np.uint64)
    ok = sharding.all_ok(ok_local, "cpu")
    if rank == 0:
        blinder = mont_array(fid, [12345])
        comm = ora.commit_scalars(fid, np.ascontiguousarray(full_chk0), blinder)
        q.put((full_val.tolist(), comm.tolist(), ok, ok_local))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n,corrupt", [(101, False), (64, True)])
def test_sharded_equals_unsharded(oracle, n, corrupt):
    from helpers import authenticated_shares, mont_array, rand_values, from_mont_array
    import pyref
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, corrupt, q)) for r in range(2)]
    for pr in procs: pr.start()
    full_val, comm, ok, ok_local0 = q.get(timeout=120)
    for pr in procs: pr.join(timeout=60)
    assert all(pr.exitcode == 0 for pr in procs)
    fid = 0; p = pyref.P[fid]
    x, y = rand_values(fid, n, 2), rand_values(fid, n, 3)
    assert from_mont_array(fid, np.array(full_val, dtype=np.uint64)) == [(u * v) % p for u, v in zip(x, y)]
    assert ok == (not corrupt)
    if corrupt:
        assert ok_local0 is True          # rank 0's shard is clean; the AND-reduce carries rank 1's failure
    # unsharded commitment over the same ordered buffer
    sharding = importlib.import_module("ark-mpc_amd.sharding")
    assert sharding.shard_sizes(n, 2) == [n // 2, n - n // 2]
    assert [sharding.shard_range(10, 4, r) for r in range(4)] == [(0, 2), (2, 5), (5, 7), (7, 10)]
    # the commitment over the gathered (ordered) MAC-check buffer equals the unsharded computation's
    k0, k1 = rand_values(fid, 2, 1); key = (k0 + k1) % p
    keys = [mont_array(fid, [k0]), mont_array(fid, [k1])]
    a, b = rand_values(fid, n, 4), rand_values(fid, n, 5)
    c = [(u * v) % p for u, v in zip(a, b)]
    sh = {nm: authenticated_shares(fid, v, key, 10 + i) for i, (nm, v) in enumerate(zip("xyabc", (x, y, a, b, c)))}
    if corrupt:
        sh["c"][0][8 * (n - 1) + 4] ^= np.uint64(1)
    de = [oracle.beaver_mask(fid, sh["x"][q_], sh["y"][q_], sh["a"][q_], sh["b"][q_]) for q_ in (0, 1)]
    opened = oracle.open_combine(fid, de[0], de[1])
    res0 = oracle.beaver_finish(fid, 0, keys[0], opened[:4 * n].copy(), opened[4 * n:].copy(), sh["a"][0], sh["b"][0], sh["c"][0])
    chk0 = oracle.mac_check_shares(fid, keys[0], np.array(full_val, dtype=np.uint64), res0)
    want = oracle.commit_scalars(fid, chk0, mont_array(fid, [12345]))
    assert comm == want.tolist()


def _msm_worker(rank, world, port, n, q):
    """MSM shards by index range: every rank folds its slice to ONE point, the world_size points are gathered in rank
    order (96 B each) and summed -- the only cross-GPU traffic of the authenticated MSM."""
    sys.path.insert(0, ROOT); sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_api
    from helpers import mont_array, rand_values
    sharding = importlib.import_module("ark-mpc_amd.sharding")
    ora = oracle_api.load()
    G = ora.g1_generator()
    P = ora.g1_batch_scalar_mul(np.tile(G, n), mont_array(0, rand_values(0, n, 21)))
    S = mont_array(0, rand_values(0, n, 22))
    lo, hi = sharding.shard_range(n, world, rank)
    part = ora.g1_msm(np.ascontiguousarray(P[12 * lo:12 * hi]), np.ascontiguousarray(S[4 * lo:4 * hi]))
    allp = sharding.gather_ordered(torch.from_numpy(part.view(np.int64).copy()), world, 12).numpy().view(np.uint64)
    if rank == 0:
        total = ora.g1_sum(np.ascontiguousarray(allp))
        want = ora.g1_msm(P, S)
        xy_a, inf_a = ora.g1_batch_to_affine(total)
        xy_b, inf_b = ora.g1_batch_to_affine(want)
        q.put((bool(np.array_equal(xy_a, xy_b) and np.array_equal(inf_a, inf_b)), int(hi - lo)))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_msm_equals_unsharded(oracle):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    n = 37
    procs = [ctx.Process(target=_msm_worker, args=(r, 2, port, n, q)) for r in range(2)]
    for pr in procs: pr.start()
    same, m0 = q.get(timeout=120)
    for pr in procs: pr.join(timeout=60)
    assert all(pr.exitcode == 0 for pr in procs)
    assert same and m0 in (18, 19)


def _ranks_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from benchlib.headline import Ranks, short_device
    r = Ranks(dist, world, rank, rank, "gloo")
    r.barrier()
    got = {"max": r.reduce(10.0 + rank, "MAX"), "min": r.reduce(10.0 + rank, "MIN"), "sum": r.reduce(1.0, "SUM"),
           "floats": r.gather_floats(0.5 * (rank + 1)),
           "objs": [short_device(o) for o in r.gather_objects({"local_device": rank, "pci_bus_id": 160 + rank, "pid": 1})]}
    r.barrier()
    q.put((rank, got))
    dist.barrier()
    dist.destroy_process_group()


def test_bench_rank_plumbing_over_gloo():
    """bench.py's process-group plumbing (benchlib/headline.py Ranks: the barrier, MAX / MIN / SUM over ranks, the per-rank float and object
    gathers behind `ms_per_step`, `ranks_seen`, `rank_devices`) with two real processes over gloo: every rank must see the same, whole picture"""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_ranks_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs: p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs: p.join(timeout=60)
    for rank in range(world):
        g = res[rank]
        assert g["max"] == 11.0 and g["min"] == 10.0 and g["sum"] == 2.0
        assert g["floats"] == [0.5, 1.0] and g["objs"] == ["0@pci160", "1@pci161"]
