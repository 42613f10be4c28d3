"""ws3d_gemm_pool (last SA layer + pool on the matrix cores) vs torch._addmm_activation + ws3d_rowmax_rows."""
import torch
from ws3d_amd import compat as c
def timeit(f, n=20):
    for _ in range(5): f()
    torch.cuda.synchronize(); a = torch.cuda.Event(True); b = torch.cuda.Event(True); a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / n
tot = [0.0, 0.0]
for (rows, ns, k, o) in [(131072, 16, 64, 128), (262144, 32, 96, 128), (32768, 16, 196, 256), (65536, 32, 196, 256),
                         (8192, 16, 256, 512), (16384, 32, 384, 512)]:
    x = torch.randn(rows, k, device="cuda"); wt = torch.randn(k, o, device="cuda") * 0.1; bias = torch.randn(o, device="cuda")
    out_a = torch.empty(rows // ns, o, device="cuda"); out_b = torch.empty_like(out_a)
    def ref():
        y = torch._addmm_activation(bias, x, wt, use_gelu=False)
        c.rowmax_rows(y, ns, out_a, 0)
    def fused():
        assert c.gemm_pool(x, wt, bias, True, ns, out_b, 0)
    ref(); fused(); torch.cuda.synchronize()
    err = float((out_a - out_b).abs().max()); scale = float(out_a.abs().max())
    ta, tb = timeit(ref), timeit(fused)
    tot[0] += ta; tot[1] += tb
    gf = 2.0 * rows * k * o / 1e9
    print(f"rows {rows:7d} ns {ns} k {k:3d} o {o:3d}: gemm+rowmax {ta:.3f} ms, fused {tb:.3f} ms ({gf / tb:.0f} TFLOP/s... GF/ms), max |diff| {err:.2e} of {scale:.1f}")
print("total: %.3f ms vs %.3f ms" % tuple(tot))
