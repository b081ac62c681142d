#!/usr/bin/env python3
"""BASELINE.json config 5 on one GPU: batch open + MAC check over BLS12-381 Fr for n shares (default 2^24), both
parties in-process.  Reports the device arithmetic (K2+K4, K5) and the commitment (K6 on the GPU + the host's
sequential SHA3 sponge) separately, as SURVEY.md section 7 asks."""
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("ark-mpc_amd")
fid = 1
n = 1 << int(os.environ.get("LOG2N", "24"))
e = pkg.Engine(fid, device=0, stream=torch.cuda.current_stream().cuda_stream)
g = torch.Generator(device="cuda"); g.manual_seed(0xA11CE005)


def rnd(cnt):
    raw = torch.randint(-(2**63), 2**63 - 1, (4 * cnt,), dtype=torch.int64, device="cuda", generator=g)
    out = torch.empty_like(raw); e.scalar_from_canonical(cnt, raw, out); return out


ks = [rnd(1), rnd(1)]
key = torch.empty_like(ks[0]); e.scalar_add(1, ks[0], ks[1], key)
keys = [k.cpu().numpy().view(np.uint64).copy() for k in ks]
v = rnd(n)
mac = torch.empty_like(v); e.scalar_mul(n, v, key.repeat(n), mac)
s0 = rnd(n); s1 = torch.empty_like(s0); e.scalar_sub(n, v, s0, s1)
m0 = rnd(n); m1 = torch.empty_like(m0); e.scalar_sub(n, mac, m0, m1)
sh = [torch.cat([s.view(n, 4), m.view(n, 4)], dim=1).contiguous().view(-1) for s, m in ((s0, m0), (s1, m1))]
del s0, s1, m0, m1, mac
mine = [torch.empty(4 * n, dtype=torch.int64, device="cuda") for _ in (0, 1)]
opened = [torch.empty(4 * n, dtype=torch.int64, device="cuda") for _ in (0, 1)]
chk = [torch.empty(4 * n, dtype=torch.int64, device="cuda") for _ in (0, 1)]
blind = [rnd(1).cpu().numpy().view(np.uint64).copy() for _ in (0, 1)]
torch.cuda.synchronize()


def device_part():
    for p in (0, 1):
        e.share_extract(n, sh[p], mine[p])
    for p in (0, 1):
        e.open_and_mac_check(n, keys[p], sh[p], mine[1 - p], opened[p], chk[p])
    return [e.mac_verify(n, chk[p], chk[1 - p]) for p in (0, 1)]


device_part(); torch.cuda.synchronize()
t0 = time.perf_counter(); reps = 5
for _ in range(reps):
    oks = device_part()
torch.cuda.synchronize()
t_dev = (time.perf_counter() - t0) / reps
assert oks == [True, True] and torch.equal(opened[0], v)
t0 = time.perf_counter()
c0 = e.commit_sha3(n, chk[0], blind[0])
t_commit = time.perf_counter() - t0
out = {"config": "2^%d shares, BLS12-381 Fr, both parties on one GPU" % int(np.log2(n)),
       "device_ms_both_parties (extract + K2+K4 + K5)": t_dev * 1e3,
       "device_shares_per_s": n / t_dev, "device_alg_GBps": 2 * n * 256 / t_dev / 1e9,
       "commit_ms_one_party_one_commitment (K6 + D2H + host SHA3)": t_commit * 1e3, "hash_MBps": 32 * n / t_commit / 1e6,
       "per_party_hashes_per_batch": 2}
print(json.dumps(out))
