"""BatchNorm2d(train)+ReLU fwd+bwd on the Stage-1 activation shapes: MIOpen vs torch's native kernels."""
import torch, torch.nn as nn
dev = "cuda"
SHAPES = [(8, 32, 4096, 16), (8, 64, 4096, 32), (8, 32, 4096, 32), (8, 128, 1024, 16), (8, 128, 1024, 32), (8, 96, 1024, 32),
          (8, 256, 256, 32), (8, 196, 256, 32), (8, 512, 64, 32), (8, 128, 16384, 1), (8, 256, 4096, 1), (8, 512, 1024, 1)]
def timeit(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize(); a = torch.cuda.Event(True); b = torch.cuda.Event(True); a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / n
tot = [0, 0]
for shp in SHAPES:
    x = torch.randn(*shp, device=dev, requires_grad=True)
    g = torch.randn(*shp, device=dev)
    bn = nn.BatchNorm2d(shp[1]).to(dev)
    def f():
        y = torch.relu_(bn(x)); y.backward(g)
    t_m = timeit(f)
    with torch.backends.cudnn.flags(enabled=False):
        t_n = timeit(f)
    gb = x.numel() * 4 / 1e9
    tot[0] += t_m; tot[1] += t_n
    print(f"{shp}: {gb*1e3:.0f} MB  miopen {t_m:.3f} ms  native {t_n:.3f} ms   (10 passes at 4 TB/s = {gb*10/4e3*1e3:.3f} ms)")
print("total", tot)
