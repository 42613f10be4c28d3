"""1x1 convolution fwd+bwd on the Stage-1 shapes: nn.Conv2d (MIOpen) vs W @ x.view(B,C,L) (hipBLASLt/rocBLAS)."""
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
    def fr():   # rows: one (8L, ci) x (ci, co) GEMM on a channels-last copy (for reference)
        xr = x.detach().view(8, ci, L).transpose(1, 2).reshape(8 * L, ci).requires_grad_(True)
        (xr @ w.t()).backward(g.view(8, co, L).transpose(1, 2).reshape(8 * L, co))
    tc, tm = timeit(fc), timeit(fm)
    tot[0] += tc; tot[1] += tm
    print(f"{ci:5d}->{co:4d} L={L:7d}: conv2d {tc:.3f} ms  matmul {tm:.3f} ms", flush=True)
print("total conv2d %.2f ms, matmul %.2f ms" % (tot[0], tot[1]))
