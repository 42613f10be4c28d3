"""A/B timing of the fused ball_query + group: fine (x, z) grid vs x slabs vs brute force.
    python scripts/ab_bq.py   ->  ms per launch incl. the binning kernel (median of 7)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ws3d_amd import compat, synth

SHAPES = [(512, 16384, 4096, 0.1, 64, 1), (256, 16384, 4096, 0.1, 64, 1), (8, 16384, 4096, 0.1, 16, 1), (8, 16384, 4096, 0.5, 32, 1),
          (8, 4096, 1024, 0.5, 16, 96), (8, 4096, 1024, 1.0, 32, 96)]
for B, N, M, r, ns, C in SHAPES:
    base = np.stack([synth.lidar_cloud(N, 200 + s) for s in range(min(B, 16))])
    pc = np.ascontiguousarray(np.tile(base, (-(-B // base.shape[0]), 1, 1))[:B])
    xyz = torch.from_numpy(pc[:, :, :3].copy()).cuda()
    feat = torch.randn((B, C, N), device="cuda")
    idx = torch.empty((B, M), dtype=torch.int32, device="cuda"); new_xyz = torch.empty((B, M, 3), device="cuda")
    compat.furthest_point_sampling_gather(B, N, M, xyz, None, idx, new_xyz)
    nbr = torch.empty((B, M, ns), dtype=torch.int32, device="cuda")
    out = torch.empty((B, 3 + C, M, ns), device="cuda")
    res = {}
    for name, mk in (("grid", lambda: compat.sort_points_x(xyz, grid=True)), ("xslab", lambda: compat.sort_points_x(xyz, grid=False)),
                     ("brute", lambda: None)):
        if name == "brute" and B > 64:
            continue
        ts = []
        for it in range(9):
            a, b, c = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            a.record(); s = mk(); b.record()
            compat.query_and_group(B, N, M, C, r, ns, True, xyz, new_xyz, feat, nbr, out, s); c.record()
            torch.cuda.synchronize()
            if it >= 2:
                ts.append((a.elapsed_time(b), b.elapsed_time(c)))
        t = np.median(np.array(ts), axis=0)
        res[name] = (t, int(nbr.sum().item()), float(out.double().sum().item()))
    ref = res["xslab"][1:]
    line = "B=%3d N=%5d M=%4d r=%.1f ns=%2d C=%3d:" % (B, N, M, r, ns, C)
    for name, (t, s1, s2) in res.items():
        line += "  %s %.3f+%.3f ms%s" % (name, t[0], t[1], "" if (s1, s2) == ref else " MISMATCH")
    gb = (M * ns * 4 + (3 + C) * N * 4 + (3 + C) * M * ns * 4 + (N + M) * 12) * B / 1e9
    line += "   [%.2f GB algorithmic -> grid %.2f TB/s]" % (gb, gb / (sum(res["grid"][0]) * 1e-3) / 1e3)
    print(line, flush=True)
