"""SharedMLP of an SA2-shaped scale over the compact (distinct) pairs vs over all m * nsample rows, from sparse to full ball-query lists
(the radius sets the fill): ws3d_pgather_gemm2(_compact) + ws3d_gemm_pool(_compact), incl. ws3d_compact_pairs."""
import numpy as np, torch
from ws3d_amd import compat as c, synth
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); a = torch.cuda.Event(True); b = torch.cuda.Event(True); a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / n * 1e3
B, N, M, ns, C, O1, O2, O3 = 8, 4096, 1024, 32, 96, 64, 96, 128
pc = torch.from_numpy(synth.make_batch("lidar", B, 16384, 5)[:, :, :3].copy()).cuda()
idx0 = torch.empty((B, N), dtype=torch.int32, device="cuda"); xyz = torch.empty((B, N, 3), device="cuda")
c.furthest_point_sampling_gather(B, 16384, N, pc, None, idx0, xyz)
idx = torch.empty((B, M), dtype=torch.int32, device="cuda"); new_xyz = torch.empty((B, M, 3), device="cuda")
c.furthest_point_sampling_gather(B, N, M, xyz, None, idx, new_xyz)
feats = torch.randn(B, N, C, device="cuda")
w1 = torch.randn(C + 3, O1, device="cuda") / C ** 0.5; b1 = torch.randn(O1, device="cuda")
w2 = torch.randn(O1, O2, device="cuda") / O1 ** 0.5; b2 = torch.randn(O2, device="cuda")
w3 = torch.randn(O2, O3, device="cuda") / O2 ** 0.5; b3 = torch.randn(O3, device="cuda")
pmat = feats.view(B * N, C) @ w1[:C]; w1x = w1[C:].contiguous()
srt = c.sort_points_x(xyz)
for r in (1.0, 3.0, 6.0, 12.0, 40.0):
    nbr = torch.zeros((B, M, ns), dtype=torch.int32, device="cuda")
    c.ball_query_wrapper(B, N, M, r, ns, new_xyz, xyz, nbr, srt)
    pairs = c.compact_pairs(nbr)
    fill = int(pairs[2].item()) / (B * M * ns)
    dense = torch.empty((B * M, O3), device="cuda"); comp = torch.zeros((B * M, O3), device="cuda")
    def run_dense():
        y = c.pgather_gemm2(pmat, 0, O1, xyz, new_xyz, nbr, w1x, b1, True, w2, b2, True); c.gemm_pool(y, w3, b3, True, ns, dense, 0)
    def run_compact():
        p = c.compact_pairs(nbr); comp.zero_()
        y = c.pgather_gemm2_compact(pmat, 0, O1, xyz, new_xyz, p, w1x, b1, True, w2, b2, True); c.gemm_pool_compact(y, p, w3, b3, comp, 0)
    run_dense(); run_compact(); torch.cuda.synchronize()
    assert torch.equal(dense, comp)
    print(f"radius {r:5.1f}: {100 * fill:5.1f} % of the rows distinct: dense {timeit(run_dense):6.1f} us, compact (incl. pair table + zeroing) {timeit(run_compact):6.1f} us, bit-identical")
