"""ws3d_gemm_pool at 1x, 4x and 16x the rows of the SA2..SA4 shapes: separates the per-launch cost (ramp-up, tail, first loads)
from the steady-state rate of the tile loop.  WS3D_GP_TILE selects the tile."""
import torch
from ws3d_amd import compat as c
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); a = torch.cuda.Event(True); b = torch.cuda.Event(True); a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / n
for (rows, ns, k, o) in [(262144, 32, 96, 128), (65536, 32, 196, 256), (16384, 32, 384, 512), (8192, 16, 256, 512)]:
    for mult in (1, 4, 16):
        r = rows * mult
        x = torch.randn(r, k, device="cuda"); wt = torch.randn(k, o, device="cuda") * 0.1; bias = torch.randn(o, device="cuda")
        out = torch.empty(r // ns, o, device="cuda")
        t = timeit(lambda: c.gemm_pool(x, wt, bias, True, ns, out, 0), n=20 if mult < 16 else 5)
        print(f"rows {r:8d} ns {ns} k {k:3d} o {o:3d}: {t * 1e3:8.1f} us  {2.0 * r * k * o / t / 1e9:6.1f} TFLOP/s")
        del x, out
