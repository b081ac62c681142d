#!/usr/bin/env python3
"""Soak of the hand-scheduled point kernels against the compiled ones: the same seeded random inputs are pushed through
PointShare x Scalar (BN254 G1 and Curve25519), the variable-base MSM and its authenticated form in two child processes --
the hand-scheduled kernels on 29-bit limbs (default), the same on 32-bit limbs (ARKMPC_EC_LIMBS=32 ARKMPC_ED_LIMBS=32), and the compiled
kernels (ARKMPC_EC_ASM=0 ARKMPC_ED_ASM=0 ARKMPC_MSM_ASM=0 ARKMPC_EDMSM_ASM=0) -- and the SHA-256 digests of the AFFINE outputs (the fixed-base generator
multiples that serve as inputs included) must agree, seed by seed.  (Jacobian / extended representatives legitimately differ between the paths.)
usage: python tools/soak_ec.py [seeds] [log2n]"""
import hashlib, importlib, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)


def child(seeds, lg):
    import numpy as np, torch
    pkg = importlib.import_module("ark-mpc_amd")
    n = 1 << lg
    res = {}
    for seed in range(seeds):
        for name, field in (("bn254", "bn254_fr"), ("curve25519", "curve25519_fr")):
            e = pkg.Engine(field, device=0, stream=torch.cuda.current_stream().cuda_stream)
            g = torch.Generator(device="cuda"); g.manual_seed(90000 + seed)
            def rnd(cnt):
                raw = torch.randint(-(2**63), 2**63 - 1, (4 * cnt,), dtype=torch.int64, device="cuda", generator=g)
                out = torch.empty_like(raw); e.scalar_from_canonical(cnt, raw, out); return out
            pw = 12 if name == "bn254" else 16
            shares = torch.empty(2 * pw * n, dtype=torch.int64, device="cuda")
            (e.scalarshare_mul_generator if name == "bn254" else e.scalarshare_mul_ed_generator)(n, rnd(2 * n), shares)
            gxy = torch.empty(8 * 2 * n, dtype=torch.int64, device="cuda")
            if name == "bn254":
                ginf = torch.empty(2 * n, dtype=torch.uint8, device="cuda"); e.g1_to_affine(2 * n, shares, gxy, ginf)
            else:
                e.ed_to_affine(2 * n, shares, gxy)
            res["%s/genmul/%d" % (name, seed)] = hashlib.sha256(gxy.cpu().numpy().tobytes()).hexdigest()
            sc = rnd(n)
            sc.view(n, 4)[: 8] = 0                                   # a few zero scalars
            out = torch.empty_like(shares)
            (e.pointshare_mul_public if name == "bn254" else e.edshare_mul_public)(n, shares, sc, out)
            xy = torch.empty(8 * 2 * n, dtype=torch.int64, device="cuda")
            if name == "bn254":
                inf = torch.empty(2 * n, dtype=torch.uint8, device="cuda")
                e.g1_to_affine(2 * n, out, xy, inf)
                h = hashlib.sha256(xy.cpu().numpy().tobytes() + inf.cpu().numpy().tobytes()).hexdigest()
            else:
                e.ed_to_affine(2 * n, out, xy)
                h = hashlib.sha256(xy.cpu().numpy().tobytes()).hexdigest()
            res["%s/smul/%d" % (name, seed)] = h
            if name == "bn254":
                pts = shares.view(2 * n, 12)[:n].contiguous().view(-1)
                scs = rnd(2 * n)
                for form, fn, cols in (("msm", e.g1_msm, 1), ("msm_auth", e.g1_msm_authenticated, 2)):
                    o = torch.empty(12 * cols, dtype=torch.int64, device="cuda")
                    fn(n, pts, scs, o)
                    a = torch.empty(8 * cols, dtype=torch.int64, device="cuda"); i2 = torch.empty(cols + 16, dtype=torch.uint8, device="cuda")
                    e.g1_to_affine(cols, o, a, i2)
                    res["%s/%s/%d" % (name, form, seed)] = hashlib.sha256(a.cpu().numpy().tobytes()).hexdigest()
            else:
                pts = shares.view(2 * n, 16)[:n].contiguous().view(-1)
                scs = rnd(2 * n)
                for form, fn, cols in (("msm", e.ed_msm, 1), ("msm_auth", e.ed_msm_authenticated, 2)):
                    o = torch.empty(16 * cols, dtype=torch.int64, device="cuda")
                    fn(n, pts, scs, o)
                    a = torch.empty(8 * cols, dtype=torch.int64, device="cuda")
                    e.ed_to_affine(cols, o, a)
                    res["%s/%s/%d" % (name, form, seed)] = hashlib.sha256(a.cpu().numpy().tobytes()).hexdigest()
            e.close()
    print(json.dumps(res))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(int(sys.argv[2]), int(sys.argv[3])); sys.exit(0)
    seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    lg = int(sys.argv[2]) if len(sys.argv) > 2 else 17
    runs = {}
    for tag, env in (("asm", {}), ("asm32", {"ARKMPC_EC_LIMBS": "32", "ARKMPC_ED_LIMBS": "32"}),
                     ("compiled", {"ARKMPC_EC_ASM": "0", "ARKMPC_ED_ASM": "0", "ARKMPC_MSM_ASM": "0", "ARKMPC_EDMSM_ASM": "0"})):
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(seeds), str(lg)], capture_output=True, text=True,
                           env=dict(os.environ, **env), timeout=3000)
        if r.returncode != 0:
            print(r.stderr[-2000:]); sys.exit(2)
        runs[tag] = json.loads(r.stdout.strip().splitlines()[-1])
    bad = [k for k in runs["asm"] if not (runs["asm"][k] == runs["compiled"][k] == runs["asm32"][k])]
    print("soak_ec: %d cases (%d seeds, 2^%d PointShares per case), %d differing between the hand-scheduled kernels (29-bit limbs, 32-bit limbs) and the compiled ones"
          % (len(runs["asm"]), seeds, lg, len(bad)))
    for k in bad: print("  DIFFERS:", k)
    sys.exit(1 if bad else 0)
