#!/usr/bin/env python3
"""Per-entry-point timings of the point-side ABI at 2^18 elements, both curves (ms per call, HIP events)."""
import importlib, json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
pkg = importlib.import_module("ark-mpc_amd")
n = 1 << int(os.environ.get("LOG2N", "18"))
def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return round(a.elapsed_time(b) / reps, 3)
for field, pw, pre, shp in (("bn254_fr", 12, "g1", "pointshare"), ("curve25519_fr", 16, "ed", "edshare")):
    e = pkg.Engine(field, device=0, stream=torch.cuda.current_stream().cuda_stream)
    g = torch.Generator(device="cuda"); g.manual_seed(3)
    def rnd(cnt):
        raw = torch.randint(-(2**63), 2**63 - 1, (4 * cnt,), dtype=torch.int64, device="cuda", generator=g)
        out = torch.empty_like(raw); e.scalar_from_canonical(cnt, raw, out); return out
    Z = lambda words: torch.empty(words, dtype=torch.int64, device="cuda")
    key = rnd(1).cpu().numpy().view(np.uint64).copy()
    ed = pre == "ed"
    ss, sc = rnd(2 * n), rnd(n)
    A, B = Z(2 * pw * n), Z(2 * pw * n)
    gen_ss = e.scalarshare_mul_ed_generator if ed else e.scalarshare_mul_generator
    gen_ss(n, ss, A); gen_ss(n, rnd(2 * n), B)
    P = A.view(2 * n, pw)[:n].contiguous().view(-1)
    out2, out1 = Z(2 * pw * n), Z(pw * n)
    f = lambda name: getattr(e, name)
    res = {"field": field, "n": n}
    res["generator_mul (n)"] = timed(lambda: f(pre + "_generator_mul")(n, sc, out1))
    res["scalar_mul (n)"] = timed(lambda: f(pre + "_scalar_mul")(n, P, sc, out1))
    res["add (n)"] = timed(lambda: f(pre + "_add")(n, P, P, out1))
    res["share_add (n shares)"] = timed(lambda: f(shp + "_add")(n, A, B, out2))
    res["share_mul_public"] = timed(lambda: f(shp + "_mul_public")(n, A, sc, out2))
    res["share_add_public"] = timed(lambda: f(shp + "_add_public")(n, 0, key, A, P, out2))
    res["scalarshare_mul_generator"] = timed(lambda: gen_ss(n, ss, out2))
    res["scalarshare_mul_point"] = timed(lambda: (e.scalarshare_mul_ed_point if ed else e.scalarshare_mul_point)(n, ss, P, out2))
    res["point_beaver_finish"] = timed(lambda: e.point_beaver_finish(n, 0, key, sc, P, ss, ss, ss, out2, ed=ed))
    res["mac_check_shares"] = timed(lambda: (e.ed_mac_check_shares if ed else e.point_mac_check_shares)(n, key, P, A, out1))
    okb = torch.empty(n + 16, dtype=torch.uint8, device="cuda")
    res["mac_verify"] = timed(lambda: (e.ed_mac_verify if ed else e.point_mac_verify)(n, P, P, okb))
    cm = Z(4 * n)
    res["commit_points_sha3"] = timed(lambda: (e.commit_ed_points_sha3 if ed else e.commit_points_sha3)(n, P, sc, cm))
    by = torch.empty(32 * n, dtype=torch.uint8, device="cuda")
    res["to_bytes"] = timed(lambda: f(pre + "_to_bytes")(n, P, by))
    res["from_bytes"] = timed(lambda: f(pre + "_from_bytes")(n, by, out1, okb))
    xy = Z(8 * n)
    res["to_affine"] = timed(lambda: (e.ed_to_affine(n, P, xy) if ed else e.g1_to_affine(n, P, xy, okb)))
    one = Z(2 * pw)
    res["sum (n points)"] = timed(lambda: f(pre + "_sum")(n, P, one))
    res["share_sum"] = timed(lambda: f(shp + "_sum")(n, A, one))
    print(json.dumps(res))
    e.close()
