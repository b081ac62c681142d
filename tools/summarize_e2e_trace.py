#!/usr/bin/env python3
"""Summary of a `rocprofv3 --kernel-trace --memory-copy-trace` run of `bench.py --only-e2e` (the streaming host-to-host path): for the LAST
one-party session in the trace (a window that starts with a 64 MiB host-to-device copy after a gap), how long each direction of the link
and the compute stream were busy, and how much of the download / kernel time ran while an upload was in flight.
    python tools/summarize_e2e_trace.py <dir with *_memory_copy_trace.csv and *_kernel_trace.csv>"""
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
if not copies:
    raise SystemExit("no memory copy trace found under " + sys.argv[1])
cs = []
for r in copies:
    d = r.get("Direction", r.get("Kind", ""))
    a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    cs.append((a, b, "up" if "HOST_TO_DEVICE" in d.upper() or "H2D" in d.upper() else ("down" if "DEVICE_TO_HOST" in d.upper() or "D2H" in d.upper() else "other"), r))
cs.sort()
# sessions: split the copy stream at gaps > 1.5 ms; keep windows whose volume looks like one party's 2^20-gate session (6 x 64 MiB up)
wins, cur = [], [cs[0]]
for c in cs[1:]:
    if c[0] - max(x[1] for x in cur) > 1_500_000:
        wins.append(cur); cur = []
    cur.append(c)
wins.append(cur)
rows = []
for w in wins:
    up = [(a, b) for a, b, k, _ in w if k == "up"]
    down = [(a, b) for a, b, k, _ in w if k == "down"]
    if not up or not down:
        continue
    t0, t1 = min(a for a, _ in up + down), max(b for _, b in up + down)
    ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in kern if t0 <= int(r["Start_Timestamp"]) <= t1]
    rows.append({"window_ms": (t1 - t0) / 1e6, "copies_up": len(up), "copies_down": len(down), "kernels": len(ks),
                 "up_busy_ms": total(union(up)) / 1e6, "down_busy_ms": total(union(down)) / 1e6, "kernel_busy_ms": total(union(ks)) / 1e6,
                 "down_under_up_ms": overlap(down, up) / 1e6, "kernels_under_up_ms": overlap(ks, up) / 1e6,
                 "up_busy_frac_of_window": total(union(up)) / (t1 - t0),
                 "down_hidden_frac": overlap(down, up) / max(1, total(union(down))), "kernels_hidden_frac": overlap(ks, up) / max(1, total(union(ks))) if ks else None,
                 "tail_after_last_upload_ms": (t1 - max(b for _, b in up)) / 1e6})
sess = [r for r in rows if r["copies_up"] >= 8 and 5.0 < r["window_ms"] < 12.0]
print(json.dumps({"windows_total": len(rows), "one_party_sessions": len(sess), "last_one_party_session": sess[-1] if sess else None,
                  "median_window_ms": sorted(r["window_ms"] for r in sess)[len(sess) // 2] if sess else None}, indent=1))
