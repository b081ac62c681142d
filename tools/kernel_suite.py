#!/usr/bin/env python3
"""Per-kernel throughput of every C-ABI entry point on the hot path (runs on the GPU box).
Prints elements/s and algorithmic GB/s (bytes each op must move once) -- the evidence table for DESIGN.md section 3."""
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


def rnd(e, cnt, g):
    raw = torch.randint(-(2**63), 2**63 - 1, (4 * cnt,), dtype=torch.int64, device="cuda", generator=g)
    out = torch.empty_like(raw); e.scalar_from_canonical(cnt, raw, out); return out


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


def main():
    fid = int(os.environ.get("FID", "0"))
    n = 1 << int(os.environ.get("LOG2N", "22"))
    e = pkg.Engine(fid, device=0, stream=torch.cuda.current_stream().cuda_stream)
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    S = lambda: rnd(e, n, g)
    sc_a, sc_b, out4 = S(), S(), torch.empty(4 * n, dtype=torch.int64, device="cuda")
    sh_a = torch.cat([S().view(n, 4), S().view(n, 4)], dim=1).contiguous().view(-1)
    sh_b = torch.cat([S().view(n, 4), S().view(n, 4)], dim=1).contiguous().view(-1)
    out8 = torch.empty(8 * n, dtype=torch.int64, device="cuda")
    out4b = torch.empty_like(out4)
    outb = torch.empty(32 * n, dtype=torch.uint8, device="cuda")
    key = rnd(e, 1, g).cpu().numpy().view(np.uint64).copy()
    rows = []
    def rec(name, fn, bytes_per_elem, reps=20):
        t = timeit(fn, reps)
        rows.append({"op": name, "elems_per_s": n / t, "alg_GBps": n * bytes_per_elem / t / 1e9, "ms": t * 1e3})
        print("%-28s %8.3f ms  %10.3e elem/s  %8.1f GB/s (alg %d B/elem)" % (name, t * 1e3, n / t, n * bytes_per_elem / t / 1e9, bytes_per_elem), flush=True)
    rec("scalar_add", lambda: e.scalar_add(n, sc_a, sc_b, out4), 96)
    rec("scalar_mul", lambda: e.scalar_mul(n, sc_a, sc_b, out4), 96)
    rec("scalar_to_bytes_be (K6)", lambda: e.scalar_to_bytes_be(n, sc_a, outb), 64)
    rec("scalar_batch_inverse", lambda: e.scalar_batch_inverse(n, sc_a, out4), 64, reps=5)
    rec("scalar_prefix_product", lambda: e.scalar_prefix_product(n, sc_a, out4), 64, reps=5)
    rec("share_add", lambda: e.share_add(n, sh_a, sh_b, out8), 192)
    rec("share_mul_public", lambda: e.share_mul_public(n, sh_a, sc_b, out8), 160)
    rec("share_add_public", lambda: e.share_add_public(n, 0, key, sh_a, sc_b, out8), 160)
    rec("open_and_mac_check (K2+K4)", lambda: e.open_and_mac_check(n, key, sh_a, sc_b, out4, out4b), 160)
    rec("mac_check_shares (K4)", lambda: e.mac_check_shares(n, key, sc_a, sh_a, out4), 128)
    sc_neg = torch.empty_like(sc_a); e.scalar_neg(n, sc_a, sc_neg)       # a verifying pair: mine + peer == 0 everywhere
    assert e.mac_verify(n, sc_a, sc_neg) is True
    rec("mac_verify (K5), blocking", lambda: e.mac_verify(n, sc_a, sc_neg), 64, reps=10)
    rec("mac_verify_async (K5)", lambda: e.mac_verify_async(n, sc_a, sc_neg), 64, reps=10)
    assert e.mac_verify_result() is True
    rec("mac_verify (K5), all elements failing", lambda: e.mac_verify(n, sc_a, sc_b), 64, reps=10)
    assert e.mac_verify(n, sc_a, sc_b) is False and e.mac_verify(n, sc_a, sc_neg) is True
    t0 = time.perf_counter(); e.commit_sha3(n, sc_a, key); t_commit = time.perf_counter() - t0
    print("commit_sha3 (K6 + host SHA3)  %8.1f ms  %6.1f MB/s hashed" % (t_commit * 1e3, 32 * n / t_commit / 1e6), flush=True)
    rows.append({"op": "commit_sha3", "ms": t_commit * 1e3, "hash_MBps": 32 * n / t_commit / 1e6})
    if fid == 0:
        m = 1 << int(os.environ.get("LOG2M", "16"))
        pts = torch.empty(12 * m, dtype=torch.int64, device="cuda")
        e.g1_generator_mul(m, sc_a[:4 * m].contiguous(), pts); torch.cuda.synchronize()
        outp = torch.empty_like(pts)
        t = timeit(lambda: e.g1_scalar_mul(m, pts, sc_b[:4 * m].contiguous(), outp), reps=3)
        print("g1_scalar_mul (K8)           %8.3f ms  %10.3e smul/s  (m = %d)" % (t * 1e3, m / t, m), flush=True)
        rows.append({"op": "g1_scalar_mul", "smul_per_s": m / t, "ms": t * 1e3, "m": m})
        t = timeit(lambda: e.g1_add(m, pts, outp, outp), reps=5)
        print("g1_add (K7)                  %8.3f ms  %10.3e add/s" % (t * 1e3, m / t), flush=True)
        rows.append({"op": "g1_add", "add_per_s": m / t, "ms": t * 1e3})
        outb32 = torch.empty(32 * m, dtype=torch.uint8, device="cuda")
        t = timeit(lambda: e.g1_to_bytes(m, outp, outb32), reps=5)
        print("g1_to_bytes                  %8.3f ms  %10.3e points/s" % (t * 1e3, m / t), flush=True)
        rows.append({"op": "g1_to_bytes", "points_per_s": m / t, "ms": t * 1e3})
        bl = rnd(e, m, g); cm = torch.empty(4 * m, dtype=torch.int64, device="cuda")
        t = timeit(lambda: e.commit_points_sha3(m, outp, bl, cm), reps=5)
        print("commit_points_sha3 (K9)      %8.3f ms  %10.3e commitments/s" % (t * 1e3, m / t), flush=True)
        rows.append({"op": "commit_points_sha3", "commitments_per_s": m / t, "ms": t * 1e3})
    print(json.dumps(rows))


if __name__ == "__main__":
    main()
