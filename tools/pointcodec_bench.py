import importlib, sys, os, json, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
pkg = importlib.import_module("ark-mpc_amd")
def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
n = 1 << 18
for field, pw, gen, tb, fb in (("bn254_fr", 12, "g1_generator_mul", "g1_to_bytes", "g1_from_bytes"), ("curve25519_fr", 16, "ed_generator_mul", "ed_to_bytes", "ed_from_bytes")):
    e = pkg.Engine(field, device=0, stream=torch.cuda.current_stream().cuda_stream)
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    raw = torch.randint(-(2**63), 2**63 - 1, (4 * n,), dtype=torch.int64, device="cuda", generator=g)
    sc = torch.empty_like(raw); e.scalar_from_canonical(n, raw, sc)
    pts = torch.empty(pw * n, dtype=torch.int64, device="cuda"); getattr(e, gen)(n, sc, pts)
    by = torch.empty(32 * n, dtype=torch.uint8, device="cuda")
    t1 = timed(lambda: getattr(e, tb)(n, pts, by))
    out = torch.empty_like(pts); ok = torch.empty(n + 16, dtype=torch.uint8, device="cuda")
    t2 = timed(lambda: getattr(e, fb)(n, by, out, ok))
    print(json.dumps({"field": field, "n": n, "to_bytes_ms": t1, "from_bytes_ms": t2, "all_ok": bool(ok[:n].all().item())}))
