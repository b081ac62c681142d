#!/usr/bin/env python3
"""Soak test of the hand-scheduled K2+K3 body: REPS launches per field/layout/party at 2^20 gates, every output word
compared with the plain C++ kernel's. Hazard mistakes in hand-written gfx950 code tend to be sporadic, hence repetition."""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
pkg = importlib.import_module("ark-mpc_amd")
REPS = int(os.environ.get("REPS", "100")); n = 1 << 20
bad = 0
for fid in (0, 1, 2, 3):
    e = pkg.Engine(fid, device=0, stream=torch.cuda.current_stream().cuda_stream)
    g = torch.Generator(device="cuda"); g.manual_seed(4242 + fid)
    def rnd(cnt):
        raw = torch.randint(-(2**63), 2**63 - 1, (4 * cnt,), dtype=torch.int64, device="cuda", generator=g)
        out = torch.empty_like(raw); e.scalar_from_canonical(cnt, raw, out); return out
    key = rnd(1).cpu().numpy().view(np.uint64).copy()
    my_de, peer_de = rnd(2 * n), rnd(2 * n)
    cols = {k: (rnd(n), rnd(n)) for k in "abc"}
    aos = {k: torch.cat([s.view(n, 4), m.view(n, 4)], dim=1).contiguous().view(-1) for k, (s, m) in cols.items()}
    opened = torch.empty_like(my_de); e.open_combine(2 * n, my_de, peer_de, opened)
    for party in (0, 1):
        ref = torch.empty(8 * n, dtype=torch.int64, device="cuda")
        e.beaver_finish(n, party, key, opened[:4 * n], opened[4 * n:], aos["a"], aos["b"], aos["c"], ref)
        rs, rm = ref.view(n, 8)[:, :4].contiguous().view(-1), ref.view(n, 8)[:, 4:].contiguous().view(-1)
        out = torch.empty(8 * n, dtype=torch.int64, device="cuda")
        o_s = torch.empty(4 * n, dtype=torch.int64, device="cuda"); o_m = torch.empty_like(o_s)
        for rep in range(REPS):
            out.zero_(); o_s.zero_(); o_m.zero_()
            e.beaver_finish_fused(n, party, key, my_de, peer_de, aos["a"], aos["b"], aos["c"], out)
            e.beaver_finish_fused_v(n, party, key, my_de, peer_de, cols["a"][0], cols["a"][1], 4, cols["b"][0], cols["b"][1], 4,
                                    cols["c"][0], cols["c"][1], 4, o_s, o_m, 4)
            torch.cuda.synchronize()
            if not (torch.equal(out, ref) and torch.equal(o_s, rs) and torch.equal(o_m, rm)):
                bad += 1
                print("MISMATCH field %d party %d rep %d" % (fid, party, rep), flush=True)
    e.close()
    print("field %d done" % fid, flush=True)
print("soak: %d mismatching launches out of %d" % (bad, 4 * 2 * REPS * 2))
sys.exit(1 if bad else 0)
