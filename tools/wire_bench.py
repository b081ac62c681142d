#!/usr/bin/env python3
"""Wire-format throughput: encode / decode of a 2^21-scalar ScalarBatch frame (the d||e exchange of a 2^20-gate batch)
on device buffers, with a CPU JSON pass over a sample for scale."""
import importlib, json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
pkg = importlib.import_module("ark-mpc_amd")
e = pkg.Engine("bn254_fr", device=0, stream=torch.cuda.current_stream().cuda_stream)
n = 1 << int(os.environ.get("LOG2N", "21"))
g = torch.Generator(device="cuda"); g.manual_seed(5)
raw = torch.randint(-(2**63), 2**63 - 1, (4 * n,), dtype=torch.int64, device="cuda", generator=g)
x = torch.empty_like(raw); e.scalar_from_canonical(n, raw, x)
cap = e.wire_frame_bound(n)
frame = torch.empty(cap, dtype=torch.uint8, device="cuda")
back = torch.empty_like(x)
ln = e.wire_encode_scalar_batch(1, n, x, frame, cap); e.wire_decode_scalar_batch(frame, ln, n, back); torch.cuda.synchronize()
def timed(fn, reps=5):
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps
te = timed(lambda: e.wire_encode_scalar_batch(1, n, x, frame, cap))
td = timed(lambda: e.wire_decode_scalar_batch(frame, ln, n, back))
# CPU: Python's json on a 2^14 sample (serde_json is faster than this; it is only a scale marker)
k = 1 << 14
canon = torch.empty_like(x); e.scalar_to_canonical(n, x, canon); torch.cuda.synchronize()
recs = canon[:4 * k].cpu().numpy().view(np.uint8).reshape(k, 32).tolist()
t0 = time.perf_counter(); txt = json.dumps({"result_id": 1, "payload": {"ScalarBatch": recs}}, separators=(",", ":")); tj = time.perf_counter() - t0
t0 = time.perf_counter(); json.loads(txt); tl = time.perf_counter() - t0
print(json.dumps({"scalars": n, "frame_bytes": ln, "encode_ms": te * 1e3, "encode_text_GBps": ln / te / 1e9, "decode_ms": td * 1e3, "decode_text_GBps": ln / td / 1e9,
                  "encode_scalars_per_s": n / te, "decode_scalars_per_s": n / td, "python_json_dumps_scalars_per_s": k / tj, "python_json_loads_scalars_per_s": k / tl}))
