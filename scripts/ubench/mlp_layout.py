"""fwd+bwd of one SharedMLP layer stack in the two layouts: Conv2d+BatchNorm2d+ReLU on (B,C,M,ns) vs
Linear+BatchNorm1d+ReLU on (B*M*ns, C) rows.  Shapes: the Stage-1 SA / FP stacks at batch 8."""
import sys, torch, torch.nn as nn, torch.nn.functional as F
dev = "cuda"
STACKS = [  # (rows_m, ns, channels...)
    ("SA1a", 8 * 4096, 16, [4, 16, 16, 32]), ("SA1b", 8 * 4096, 32, [4, 32, 32, 64]),
    ("SA2a", 8 * 1024, 16, [99, 64, 64, 128]), ("SA2b", 8 * 1024, 32, [99, 64, 96, 128]),
    ("SA3a", 8 * 256, 16, [259, 128, 196, 256]), ("SA3b", 8 * 256, 32, [259, 128, 196, 256]),
    ("SA4a", 8 * 64, 16, [515, 256, 256, 512]), ("SA4b", 8 * 64, 32, [515, 256, 384, 512]),
    ("FP1", 8 * 16384, 1, [257, 128, 128]), ("FP2", 8 * 4096, 1, [608, 256, 256]),
    ("FP3", 8 * 1024, 1, [768, 512, 512]), ("FP4", 8 * 256, 1, [1536, 512, 512]),
]
def timeit(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize(); a = torch.cuda.Event(True); b = torch.cuda.Event(True); a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / n
tot = [0.0, 0.0]
for name, m, ns, ch in STACKS:
    conv = nn.Sequential(*[x for i in range(len(ch) - 1) for x in (nn.Conv2d(ch[i], ch[i + 1], 1, bias=False), nn.BatchNorm2d(ch[i + 1]), nn.ReLU(True))]).to(dev)
    lin = nn.Sequential(*[x for i in range(len(ch) - 1) for x in (nn.Linear(ch[i], ch[i + 1], bias=False), nn.BatchNorm1d(ch[i + 1]), nn.ReLU(True))]).to(dev)
    x4 = torch.randn(8, ch[0], m // 8, ns, device=dev, requires_grad=True)
    x2 = torch.randn(m * ns, ch[0], device=dev, requires_grad=True)
    def f4():
        y = conv(x4); y.sum().backward()
    def f2():
        y = lin(x2); y.sum().backward()
    t4, t2 = timeit(f4), timeit(f2)
    tot[0] += t4; tot[1] += t2
    print(f"{name}: rows {m*ns} ch {ch}: conv2d/NCHW {t4:.3f} ms, linear/rows {t2:.3f} ms", flush=True)
print(f"total: NCHW {tot[0]:.2f} ms, rows {tot[1]:.2f} ms")
