"""How long does the host need to ISSUE one eager Stage-1 step (forward + proposals + roipool3d, batch 8), compared with the
time the GPU needs to run it?  (eager latency mode is host-bound if the first exceeds the second)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench_c3

wl = bench_c3.C3(8, 0, 1, depth=1)
for _ in range(3):
    wl.step(eager=True)
torch.cuda.synchronize()
issue, total = [], []
for _ in range(20):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    wl.step(eager=True)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    issue.append((t1 - t0) * 1e3); total.append((t2 - t0) * 1e3)
print("host issue %.3f ms (median), issue + wait %.3f ms" % (float(np.median(issue)), float(np.median(total))))
