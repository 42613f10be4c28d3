"""ws3d_conv1x1_wgrad vs the library's convolution weight gradient on the Stage-1 layer shapes (batch 8)."""
import torch
from ws3d_amd import compat as c
LAYERS = [(4, 16, 4096 * 16), (16, 16, 4096 * 16), (16, 32, 4096 * 16), (4, 32, 4096 * 32), (32, 32, 4096 * 32), (32, 64, 4096 * 32),
          (99, 64, 1024 * 16), (64, 64, 1024 * 16), (64, 128, 1024 * 16), (99, 64, 1024 * 32), (64, 96, 1024 * 32), (96, 128, 1024 * 32),
          (259, 128, 256 * 16), (128, 196, 256 * 16), (196, 256, 256 * 16), (259, 128, 256 * 32), (128, 196, 256 * 32), (196, 256, 256 * 32),
          (515, 256, 64 * 16), (256, 256, 64 * 16), (256, 512, 64 * 16), (515, 256, 64 * 32), (256, 384, 64 * 32), (384, 512, 64 * 32),
          (257, 128, 16384), (128, 128, 16384), (608, 256, 4096), (256, 256, 4096), (768, 512, 1024), (512, 512, 1024), (1536, 512, 256), (512, 512, 256),
          (128, 128, 16384), (128, 128, 16384), (128, 1, 16384), (128, 40, 16384)]
def timeit(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize(); a = torch.cuda.Event(True); b = torch.cuda.Event(True); a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / n
tot = [0.0, 0.0]; worst = 0.0
for ci, co, L in LAYERS:
    x = torch.randn(8, ci, L, 1, device="cuda"); g = torch.randn(8, co, L, 1, device="cuda")
    w = torch.randn(co, ci, 1, 1, device="cuda")
    def lib():
        return torch.ops.aten.convolution_backward(g, x, w, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, [False, True, False])[1]
    def own():
        return c.conv1x1_wgrad(g, x)
    ref = torch.einsum("bol,bcl->oc", g.view(8, co, L).double(), x.view(8, ci, L).double())
    e_own = float((own().double() - ref).abs().max() / ref.abs().max()); e_lib = float((lib().view(co, ci).double() - ref).abs().max() / ref.abs().max())
    worst = max(worst, e_own)
    tl, to = timeit(lib), timeit(own)
    tot[0] += tl; tot[1] += to
    print(f"{ci:5d}->{co:4d} L={L:7d}: library {tl:.3f} ms (err {e_lib:.1e})  ws3d {to:.3f} ms (err {e_own:.1e})", flush=True)
print("total: library %.2f ms, ws3d %.2f ms; worst relative error %.1e" % (tot[0], tot[1], worst))
