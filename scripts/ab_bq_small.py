"""small-cloud ball query + group (SA3 / SA4 shapes, channels-last): binned (grid) vs brute force"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ws3d_amd import compat, synth
for (B, N, M, r, ns, C) in [(8, 1024, 256, 1.0, 16, 256), (8, 1024, 256, 2.0, 32, 256), (8, 256, 64, 2.0, 16, 512), (8, 256, 64, 4.0, 32, 512)]:
    pc = np.stack([synth.lidar_cloud(16384, 300 + s)[:N] for s in range(B)])
    xyz = torch.from_numpy(pc[:, :, :3].copy()).cuda()
    feat = torch.randn((B, N, C), device="cuda")
    idx = torch.empty((B, M), dtype=torch.int32, device="cuda"); new_xyz = torch.empty((B, M, 3), device="cuda")
    compat.furthest_point_sampling_gather(B, N, M, xyz, None, idx, new_xyz)
    line = "B=%d N=%d M=%d r=%.1f ns=%d C=%d:" % (B, N, M, r, ns, C)
    ref = None
    for name in ("brute", "grid"):
        ts = []
        for it in range(9):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            s = compat.sort_points_x(xyz, min_n=1, grid=True) if name == "grid" else None
            g = compat.query_and_group_nlc(r, ns, xyz, new_xyz, feat, True, s)
            b.record(); torch.cuda.synchronize()
            if it >= 2: ts.append(a.elapsed_time(b))
        chk = float(g.double().sum().item())
        ref = chk if ref is None else ref
        line += "  %s %.3f ms%s" % (name, float(np.median(ts)), "" if chk == ref else " MISMATCH")
    print(line, flush=True)
