#!/usr/bin/env python3
"""BASELINE.json config 5 across the GPUs of one node: 2^LOG2N authenticated shares over BLS12-381 Fr (default 2^24),
batch open + MAC check sharded by contiguous index range, one process per GPU:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P tools/config5_dist.py

Every rank runs extract / open_and_mac_check / mac_verify on ITS slice only (no data-path collective).  The one exchange
the path really has follows: the opened values and the MAC-check shares (32 B per element each) are
all-gathered over RCCL in rank order, so that rank 0 holds the ordered byte stream the sequential SHA3 commitment is
defined over (commitment.rs:30-43) and the caller-visible result is contiguous; the verify flag is AND-reduced.
The shares are generated from the GLOBAL element index, so the gathered result and the commitment do not depend on the
number of ranks -- which the script checks against a single-slice run of the same code on rank 0 for sizes <= 2^22."""
import importlib
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("ark-mpc_amd")
sharding = importlib.import_module("ark-mpc_amd.sharding")
FID = 1


def splitmix(idx, salt):
    """counter-based 64-bit words from the global element index (int64 tensors, wrap-around arithmetic)"""
    z = idx * (-7046029254386353131) + salt                      # 0x9E3779B97F4A7C15 as int64
    z = (z ^ (z >> 30)) * (-4658895280553007687)                 # 0xBF58476D1CE4E5B9
    z = (z ^ (z >> 27)) * (-7723592293110705685)                 # 0x94D049BB133111EB
    return z ^ (z >> 31)


def field_elems(e, lo, cnt, salt):
    idx = torch.arange(lo, lo + cnt, dtype=torch.int64, device="cuda")
    raw = torch.stack([splitmix(idx, salt + k) for k in range(4)], dim=1).contiguous().view(-1)
    out = torch.empty_like(raw); e.scalar_from_canonical(cnt, raw, out); return out      # 256-bit words reduced mod p


def run_slice(e, lo, cnt, keys, key):
    """both parties' open + MAC check on elements [lo, lo + cnt) -> (opened, chk0, chk1, ok, seconds)"""
    v = field_elems(e, lo, cnt, 11)
    mac = torch.empty_like(v); e.scalar_mul(cnt, v, key.repeat(cnt), mac)
    s0 = field_elems(e, lo, cnt, 23); s1 = torch.empty_like(s0); e.scalar_sub(cnt, v, s0, s1)
    m0 = field_elems(e, lo, cnt, 37); m1 = torch.empty_like(m0); e.scalar_sub(cnt, mac, m0, m1)
    sh = [torch.cat([s.view(cnt, 4), m.view(cnt, 4)], dim=1).contiguous().view(-1) for s, m in ((s0, m0), (s1, m1))]
    mine = [torch.empty(4 * cnt, dtype=torch.int64, device="cuda") for _ in (0, 1)]
    opened = [torch.empty(4 * cnt, dtype=torch.int64, device="cuda") for _ in (0, 1)]
    chk = [torch.empty(4 * cnt, dtype=torch.int64, device="cuda") for _ in (0, 1)]
    def go():
        for p in (0, 1): e.share_extract(cnt, sh[p], mine[p])
        for p in (0, 1): e.open_and_mac_check(cnt, keys[p], sh[p], mine[1 - p], opened[p], chk[p])
        return all(e.mac_verify(cnt, chk[p], chk[1 - p]) for p in (0, 1))
    go(); torch.cuda.synchronize()
    t0 = time.perf_counter(); ok = go(); torch.cuda.synchronize(); t = time.perf_counter() - t0
    assert torch.equal(opened[0], v) and torch.equal(opened[1], v)
    return opened[0], chk[0], chk[1], ok, t


def main():
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("DIST_BACKEND", "nccl")          # gloo: lets a test oversubscribe one GPU with several ranks
    local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    use_dist = "TORCHELASTIC_RUN_ID" in os.environ or world > 1
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend)
    n = 1 << int(os.environ.get("LOG2N", "24"))
    e = pkg.Engine(FID, device=local, stream=torch.cuda.current_stream().cuda_stream)
    ks = [field_elems(e, 0, 1, 101), field_elems(e, 0, 1, 103)]             # identical on every rank
    key = torch.empty_like(ks[0]); e.scalar_add(1, ks[0], ks[1], key)
    keys = [k.cpu().numpy().view(np.uint64).copy() for k in ks]
    blind = field_elems(e, 0, 1, 107).cpu().numpy().view(np.uint64).copy()
    lo, hi = sharding.shard_range(n, world, rank)
    opened, chk0, chk1, ok_local, t_dev = run_slice(e, lo, hi - lo, keys, key)
    if use_dist:
        dist.barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        if backend == "nccl":
            full_opened = sharding.gather_ordered(opened, n, 4)
            full_chk0 = sharding.gather_ordered(chk0, n, 4)
            ok = sharding.all_ok(ok_local, "cuda")
        else:                                                # gloo collectives run on host tensors
            full_opened = sharding.gather_ordered(opened.cpu(), n, 4).cuda()
            full_chk0 = sharding.gather_ordered(chk0.cpu(), n, 4).cuda()
            ok = sharding.all_ok(ok_local, "cpu")
        torch.cuda.synchronize(); t_gather = time.perf_counter() - t0
    else:
        full_opened, full_chk0, ok, t_gather = opened, chk0, ok_local, 0.0
    out = None
    if rank == 0:
        t0 = time.perf_counter()
        comm = e.commit_sha3(n, full_chk0, blind)                # one sequential sponge over the ordered stream
        t_hash = time.perf_counter() - t0
        out = {"config": "2^%d shares, BLS12-381 Fr, open + MAC check" % int(np.log2(n)), "n_gpus": world, "verify_ok": bool(ok),
               "device_ms_per_rank (both parties, this rank's slice)": t_dev * 1e3, "gather_ms (opened + chk, RCCL all-gather)": t_gather * 1e3,
               "commit_ms (rank 0, host SHA3 over the gathered stream)": t_hash * 1e3,
               "commitment": [int(x) for x in np.asarray(comm).view(np.uint64)]}
        if n <= (1 << 22) and world > 0:
            # sharding invariance: the same code over the whole range in one slice
            o1, c01, _, ok1, _ = run_slice(e, 0, n, keys, key)
            same = torch.equal(o1, full_opened) and torch.equal(c01, full_chk0) and ok1 == ok
            out["equals_single_slice_run"] = bool(same)
            assert same
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if out is not None:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
