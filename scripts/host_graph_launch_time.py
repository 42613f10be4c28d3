"""Host time of ONE hipGraph replay of the Stage-1 step (the throughput mode issues one per batch): is the launching thread a
bound at ~1 ms per batch?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench_c3

wl = bench_c3.C3(8, 0, 1, depth=4)
assert wl.capture()
for _ in range(8):
    wl.step()
torch.cuda.synchronize()
ts = []
for _ in range(40):
    t0 = time.perf_counter()
    wl.step()
    ts.append((time.perf_counter() - t0) * 1e3)
torch.cuda.synchronize()
print("host time per submitted batch (graph replay + exchange bookkeeping): median %.3f ms, min %.3f, max %.3f" % (float(np.median(ts)), min(ts), max(ts)))
