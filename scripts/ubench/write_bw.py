"""Pure-write and copy bandwidth of the box (torch fill_ / copy_) -- the ceiling of write-dominated kernels (grouping, roipool)."""
import torch, numpy as np
for gb in (0.5, 2.7):
    n = int(gb * 1e9 / 4)
    a = torch.empty(n, device="cuda"); b = torch.empty(n, device="cuda")
    for name, fn, byt in (("fill_", lambda: a.fill_(1.0), 4 * n), ("zero_", lambda: a.zero_(), 4 * n), ("copy_", lambda: b.copy_(a), 8 * n)):
        ts = []
        for it in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
        t = float(np.median(ts))
        print("%.1f GB %-6s %.3f ms  %.2f TB/s (read+written bytes)" % (gb, name, t, byt / t / 1e9))
