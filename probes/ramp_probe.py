import importlib, sys, os, time, json
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from benchlib import common as bench
pkg = importlib.import_module("ark-mpc_amd")
torch.cuda.set_device(0)
eng = pkg.Engine(0, device=0, stream=torch.cuda.current_stream().cuda_stream)
n = 1 << 20
sets = [bench.build_workload(eng, n, seed=0xA11CE002 + 7919 * k, layout="split") for k in range(2)]
calls = [bench.prepare_step(eng, n, ps, "split", 1, "01") for ps, _ in sets]
def run(warm, steps, pre_idle_ms=0):
    torch.cuda.synchronize()
    if pre_idle_ms: time.sleep(pre_idle_ms / 1e3)
    for w in range(warm): bench.step(calls[w % 2])
    torch.cuda.synchronize()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    evs[0].record()
    for s in range(steps):
        bench.step(calls[s % 2]); evs[s + 1].record()
    torch.cuda.synchronize()
    return [evs[i].elapsed_time(evs[i + 1]) for i in range(steps)]
for warm, idle in ((5, 0), (5, 50), (50, 0), (0, 0), (5, 0)):
    t = run(warm, 20, idle)
    print("warm", warm, "idle_ms", idle, "steps ms:", " ".join("%.3f" % x for x in t), " total %.3f" % sum(t))
