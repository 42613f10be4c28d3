"""SharedMLP of an SA2-shaped scale over the compact (distinct) pairs vs over all m * nsample rows, from sparse to full ball-query lists
(the radius sets the fill): ws3d_pgather_gemm2(_compact) + ws3d_gemm_pool(_compact), incl. the pair table -- and the device-side
dispatch of round 3: both forms launched with their launch gates, the pair total of the batch decides in the kernels' prologues."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
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
    disp = torch.zeros((B * M, O3), device="cuda")
    limit = int(0.55 * B * M * ns)
    def run_dispatch():     # what fastpath.sa_forward issues: lists + pair table in one launch, then both forms, gated
        nb, p = c.ball_query_pairs(r, ns, xyz, new_xyz, srt); disp.zero_()
        y = c.pgather_gemm2_compact(pmat, 0, O1, xyz, new_xyz, p, w1x, b1, True, w2, b2, True, limit=limit); c.gemm_pool_compact(y, p, w3, b3, disp, 0, limit=limit)
        c.pgather_gemm2(pmat, 0, O1, xyz, new_xyz, nb, w1x, b1, True, w2, b2, True, out=y, gate=(p[2], limit)); c.gemm_pool(y, w3, b3, True, ns, disp, 0, gate=(p[2], limit))
    def run_lists():        # the search alone (its time is inside run_dispatch, not inside the two columns before it)
        c.ball_query_pairs(r, ns, xyz, new_xyz, srt)
    run_dense(); run_compact(); run_dispatch(); torch.cuda.synchronize()
    assert torch.equal(dense, comp) and torch.equal(dense, disp)
    td, tc, tx, tl = timeit(run_dense), timeit(run_compact), timeit(run_dispatch), timeit(run_lists)
    print(f"radius {r:5.1f}: {100 * fill:5.1f} % of the rows distinct: dense {td:6.1f} us, compact (incl. pair table + zeroing) {tc:6.1f} us, "
          f"device-side dispatch {tx - tl:6.1f} us (+ {tl:5.1f} us of search with the pair table; runs the {'compact' if fill <= 0.55 else 'dense'} form), bit-identical")
