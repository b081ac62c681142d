#!/usr/bin/env python3
"""Summary of a `rocprofv3 --kernel-trace --memory-copy-trace` run of `bench.py --only-e2e` (the streaming host-to-host path): for the LAST
one-party session in the trace (a window that starts with a 64 MiB host-to-device copy after a gap), how long each direction of the link
and the compute stream were busy, and how much of the download / kernel time ran while an upload was in flight.
Sessions on vectors the caller pinned run as two zero-copy kernels instead (no copies): their durations and link rates are summarised too.
    python tools/summarize_e2e_trace.py <dir with *_memory_copy_trace.csv and *_kernel_trace.csv> [log2n = 20]"""
import csv
import glob
import json
import os
import sys


def load(pattern):
    f = glob.glob(os.path.join(sys.argv[1], "**", pattern), recursive=True)
    if not f:
        return []
    return list(csv.DictReader(open(f[0])))


def union(iv):
    iv = sorted(iv)
    out = []
    for a, b in iv:
        if out and a <= out[-1][1]:
            out[-1][1] = max(out[-1][1], b)
        else:
            out.append([a, b])
    return out


def total(iv):
    return sum(b - a for a, b in iv)


def overlap(x, y):
    x, y = union(x), union(y)
    i = j = 0
    t = 0
    while i < len(x) and j < len(y):
        a, b = max(x[i][0], y[j][0]), min(x[i][1], y[j][1])
        if b > a:
            t += b - a
        if x[i][1] < y[j][1]:
            i += 1
        else:
            j += 1
    return t


copies = load("*memory_copy_trace.csv")
kern = load("*kernel_trace.csv")
if not copies and not kern:
    raise SystemExit("no memory copy / kernel trace found under " + sys.argv[1])
cs = []
for r in copies:
    d = r.get("Direction", r.get("Kind", ""))
    a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    cs.append((a, b, "up" if "HOST_TO_DEVICE" in d.upper() or "H2D" in d.upper() else ("down" if "DEVICE_TO_HOST" in d.upper() or "D2H" in d.upper() else "other"), r))
cs.sort()
# one party's session = 22 host-to-device copies (x, y, a, six chunks of b, c, twelve chunks of the peer's d||e).  The bench's e2e leg ends each
# buffer mode with four ISOLATED sessions (4 ms of idle link before each): those are the windows of exactly 22 uploads between gaps > 2 ms
ups = [c for c in cs if c[2] == "up"]
wins, cur = [], ups[:1]
for c in ups[1:]:
    if c[0] - max(x[1] for x in cur) > 2_000_000:
        wins.append(cur); cur = []
    cur.append(c)
wins.append(cur)
rows = []
for w in wins:
    if len(w) != 22:
        continue
    up = [(a, b) for a, b, _, _ in w]
    t0, t_up_end = up[0][0], up[-1][1]
    # downloads: device-to-host copies in the copy trace and the runtime's blit kernels (__amd_rocclr_copyBuffer: this ROCm moves D2H to pinned memory
    # with a shader copy); kernels: everything else that started inside the session
    down = [(a, b) for a, b, k, _ in cs if k == "down" and t0 <= a <= t_up_end + 2_000_000]
    ks, blits = [], []
    for r in kern:
        a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if t0 <= a <= t_up_end + 2_000_000:
            (blits if "copyBuffer" in r["Kernel_Name"] else ks).append((a, b))
    down += blits
    t1 = max([t_up_end] + [b for _, b in down])
    rows.append({"window_ms": (t1 - t0) / 1e6, "copies_up": len(up), "downloads": len(down), "kernels": len(ks),
                 "up_busy_ms": total(union(up)) / 1e6, "down_busy_ms": total(union(down)) / 1e6, "kernel_busy_ms": total(union(ks)) / 1e6,
                 "up_busy_frac_of_window": total(union(up)) / (t1 - t0),
                 "down_hidden_frac": overlap(down, up) / max(1, total(union(down))), "kernels_hidden_frac": overlap(ks, up) / max(1, total(union(ks))) if ks else None,
                 "tail_after_last_upload_ms": (t1 - t_up_end) / 1e6,
                 "first_two_uploads_GBps": [round(64 * 1.048576e6 / ((b - a) / 1e9) / 1e9, 1) for a, b in up[:2]], "third_upload_GBps": round(64 * 1.048576e6 / ((up[2][1] - up[2][0]) / 1e9) / 1e9, 1)})
# zero-copy sessions (vectors pinned by the caller): no copies at all -- one k_hostmul_mask and one k_hostmul_finish per 2^20 gates that read and
# write the host records in place.  One-party sessions = a mask kernel followed by its finish kernel with no other session's kernel in between.
log2n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
hk = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "mask" if "hostmul_mask" in r["Kernel_Name"] else "finish")
            for r in kern if "k_hostmul_" in r["Kernel_Name"])
zc = []
for i in range(len(hk) - 1):
    (a0, b0, k0), (a1, b1, k1) = hk[i], hk[i + 1]
    alone = (i == 0 or hk[i - 1][1] <= a0) and (i + 2 >= len(hk) or hk[i + 2][0] >= b1)
    if k0 == "mask" and k1 == "finish" and b0 <= a1 and a1 - b0 < 1_000_000 and alone:
        n = 1 << log2n
        zc.append({"mask_ms": (b0 - a0) / 1e6, "finish_ms": (b1 - a1) / 1e6, "gap_between_phases_ms": (a1 - b0) / 1e6, "window_ms": (b1 - a0) / 1e6,
                   "mask_link_GBps_up": 256 * n / (b0 - a0), "finish_link_GBps_up": 128 * n / (b1 - a1), "session_link_GBps_up": 384 * n / (b1 - a0)})
med = lambda key: sorted(z[key] for z in zc)[len(zc) // 2] if zc else None
print(json.dumps({"sessions_found": len(rows), "last_one_party_session": rows[-1] if rows else None,
                  "median_window_ms": sorted(r["window_ms"] for r in rows)[len(rows) // 2] if rows else None,
                  "zero_copy_one_party_sessions": {"found": len(zc), "log2n": log2n,
                                                   "median": {k: med(k) for k in ("mask_ms", "finish_ms", "gap_between_phases_ms", "window_ms", "mask_link_GBps_up",
                                                                                  "finish_link_GBps_up", "session_link_GBps_up")} if zc else None,
                                                   "what": "k_hostmul_mask reads 256 B per gate over the link (x, y, a, b records) and writes 64 B back (d||e); k_hostmul_finish "
                                                           "reads 128 B (c, the peer's d||e) and writes 64 B (the result record); GB/s = the READ direction only"}}, indent=1))
