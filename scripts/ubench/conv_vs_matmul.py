"""1x1 convolution fwd+bwd on the Stage-1 shapes: nn.Conv2d (MIOpen) vs W @ x.view(B,C,L) (rocBLAS batched)
vs F.linear on a rows (B*L, C) tensor."""
import torch, torch.nn as nn
dev = "cuda"
LAYERS = [(4, 16, 4096 * 16), (16, 16, 4096 * 16), (16, 32, 4096 * 16), (4, 32, 4096 * 32), (32, 32, 4096 * 32), (32, 64, 4096 * 32),
          (99, 64, 1024 * 16), (64, 64, 1024 * 16), (64, 128, 1024 * 16), (99, 64, 1024 * 32), (64, 96, 1024 * 32), (96, 128, 1024 * 32),
          (259, 128, 256 * 16), (128, 196, 256 * 16), (196, 256, 256 * 16), (259, 128, 256 * 32), (128, 196, 256 * 32), (196, 256, 256 * 32),
          (515, 256, 64 * 16), (256, 256, 64 * 16), (256, 512, 64 * 16), (515, 256, 64 * 32), (256, 384, 64 * 32), (384, 512, 64 * 32),
          (257, 128, 16384), (128, 128, 16384), (608, 256, 4096), (256, 256, 4096), (768, 512, 1024), (512, 512, 1024), (1536, 512, 256), (512, 512, 256)]
def timeit(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize(); a = torch.cuda.Event(True); b = torch.cuda.Event(True); a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / n
tot = [0.0, 0.0, 0.0]
for ci, co, L in LAYERS:
    x = torch.randn(8, ci, L, 1, device=dev, requires_grad=True)
    g = torch.randn(8, co, L, 1, device=dev)
    conv = nn.Conv2d(ci, co, 1, bias=False).to(dev)
    w = conv.weight.detach().view(co, ci).clone().requires_grad_(True)
    def fc():
        conv(x).backward(g)
    def fm():
        torch.matmul(w, x.view(8, ci, L)).backward(g.view(8, co, L))
    xr = torch.randn(8 * L, ci, device=dev, requires_grad=True)     # rows layout: one (8L, ci) x (ci, co) GEMM
    gr = torch.randn(8 * L, co, device=dev)
    def fr():
        torch.nn.functional.linear(xr, w).backward(gr)
    tc, tm, tr = timeit(fc), timeit(fm), timeit(fr)
    tot[0] += tc; tot[1] += tm; tot[2] += tr
    print(f"{ci:5d}->{co:4d} L={L:7d}: conv2d {tc:.3f} ms  matmul {tm:.3f} ms  rows-linear {tr:.3f} ms", flush=True)
print("total conv2d %.2f ms, matmul %.2f ms, rows-linear %.2f ms" % tuple(tot))
