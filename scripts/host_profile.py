"""cProfile of the host side of one eager Stage-1 step (forward + proposals + roipool3d, batch 8): where the launching thread spends its time"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench_c3

wl = bench_c3.C3(8, 0, 1, depth=1)
for _ in range(5):
    wl.step(eager=True)
torch.cuda.synchronize()
issue, total = [], []
for _ in range(20):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); wl.step(eager=True); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    issue.append((t1 - t0) * 1e3); total.append((t2 - t0) * 1e3)
print("host issue %.3f ms (median), issue + wait %.3f ms" % (float(np.median(issue)), float(np.median(total))))
pr = cProfile.Profile()
for _ in range(10):
    torch.cuda.synchronize()
    pr.enable(); wl.step(eager=True); pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumtime").print_stats(45)
