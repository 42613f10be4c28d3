"""ws3d_gather_gemm3_pool (grouping + 3 layers + pool in one kernel) against ws3d_gather_gemm2 + ws3d_gemm_pool on the SA2 / SA3
shapes of the c3 network at batch 8."""
import numpy as np, torch
from ws3d_amd import compat as c, synth
def timeit(f, n=30):
    for _ in range(5): f()
    torch.cuda.synchronize(); a = torch.cuda.Event(True); b = torch.cuda.Event(True); a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / n * 1e3
pc = torch.from_numpy(synth.make_batch("lidar", 8, 16384, 5)[:, :, :3].copy()).cuda()
for N, M, ns, C, O1, O2, O3, r in [(4096, 1024, 16, 96, 64, 64, 128, 0.5), (4096, 1024, 32, 96, 64, 96, 128, 1.0),
                                    (1024, 256, 16, 256, 128, 196, 256, 1.0), (1024, 256, 32, 256, 128, 196, 256, 2.0)]:
    B = 8
    xyz = pc[:, :N].contiguous(); feats = torch.randn(B, N, C, device="cuda")
    idx = torch.empty((B, M), dtype=torch.int32, device="cuda"); new_xyz = torch.empty((B, M, 3), device="cuda")
    c.furthest_point_sampling_gather(B, N, M, xyz, None, idx, new_xyz)
    nbr = torch.zeros((B, M, ns), dtype=torch.int32, device="cuda")
    c.ball_query_wrapper(B, N, M, r, ns, new_xyz, xyz, nbr, c.sort_points_x(xyz))
    w1 = torch.randn(C + 3, O1, device="cuda") / C ** 0.5; b1 = torch.randn(O1, device="cuda")
    w2 = torch.randn(O1, O2, device="cuda") / O1 ** 0.5; b2 = torch.randn(O2, device="cuda")
    w3 = torch.randn(O2, O3, device="cuda") / O2 ** 0.5; b3 = torch.randn(O3, device="cuda")
    out = torch.empty(B * M, O3, device="cuda")
    def two():
        y = c.gather_gemm2(feats, xyz, new_xyz, nbr, w1, b1, True, w2, b2, True); c.gemm_pool(y, w3, b3, True, ns, out, 0)
    def one(): assert c.gather_gemm3_pool(feats, xyz, new_xyz, nbr, w1, b1, True, w2, b2, True, w3, b3, True, out, 0)
    gf = 2.0 * B * M * ns * ((C + 3) * O1 + O1 * O2 + O2 * O3) / 1e9
    t2, t1 = timeit(two), timeit(one)
    print(f"rows {B * M * ns} {C + 3}->{O1}->{O2}->{O3} ns {ns}: two kernels {t2:.1f} us ({gf / t2 * 1e3:.0f} TFLOP/s), one kernel {t1:.1f} us ({gf / t1 * 1e3:.0f} TFLOP/s)")
