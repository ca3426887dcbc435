"""Config 5: how long the HOST takes to enqueue one step (no synchronisation inside the loop) against the device time per
step -- whether the Conformer step is launch-bound.  usage: python tools/host_ahead_c5.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
setup = open(os.path.join(ROOT, "tools", "conformer_bench.py")).read().split("n = int(os.environ.get")[0]
g = {"__name__": "setup", "__file__": os.path.join(ROOT, "tools", "conformer_bench.py")}
exec(compile(setup, "conformer_bench_setup", "exec"), g)
step = g["step"]
for _ in range(3):
    step()
torch.cuda.synchronize()
ts = [time.perf_counter()]
for _ in range(6):
    step()
    ts.append(time.perf_counter())
host = [(b - a) * 1e3 for a, b in zip(ts, ts[1:])]
torch.cuda.synchronize()
tot = (time.perf_counter() - ts[0]) * 1e3 / 6
print("host enqueue ms per step:", [round(h, 1) for h in host], " wall per step incl. drain:", round(tot, 1))
