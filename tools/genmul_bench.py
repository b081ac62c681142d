#!/usr/bin/env python3
"""Generator multiplication (batch_mul_generator): fixed-base table path vs the variable-base GLV path (ARKMPC_NO_FIXED_BASE=1)."""
import importlib, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    pkg = importlib.import_module("ark-mpc_amd")
    e = pkg.Engine("bn254_fr", device=0, stream=torch.cuda.current_stream().cuda_stream)
    n = 1 << 18
    g = torch.Generator(device="cuda"); g.manual_seed(3)
    raw = torch.randint(-(2**63), 2**63 - 1, (8 * n,), dtype=torch.int64, device="cuda", generator=g)
    ss = torch.empty_like(raw); e.scalar_from_canonical(2 * n, raw, ss)
    out = torch.empty(24 * n, dtype=torch.int64, device="cuda")
    e.scalarshare_mul_generator(n, ss, out); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): e.scalarshare_mul_generator(n, ss, out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(json.dumps({"op": "2^18 ScalarShare x generator = 2^19 generator muls", "fixed_base": os.environ.get("ARKMPC_NO_FIXED_BASE") != "1",
                      "ms": ms, "generator_muls_per_s": 2 * n / ms * 1e3}))
else:
    for env in ({}, {"ARKMPC_NO_FIXED_BASE": "1"}):
        print(subprocess.run([sys.executable, __file__, "child"], env={**os.environ, **env}, capture_output=True, text=True).stdout.strip())
