"""the searches of one c3 batch as fastpath issues them (ball_query_pairs: lists + compact pairs per scale, levels 1-4 of 8 hdl64 scenes),
each timed alone: what the geometry chain of the eager pass waits for.    python scripts/ubench/bq_pairs_levels.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from ws3d_amd import compat as c, synth, stage1, pn2_ops

cfg = stage1.DEFAULT_CFG
for kind in ("hdl64", "lidar"):
    B = 8
    xyz = torch.from_numpy(np.stack([synth.cloud(kind, 16384, 2000 + s)[:, :3] for s in range(B)])).cuda()
    line = kind + ":"
    for lvl, (m, radii, nss) in enumerate(zip(cfg.npoints, cfg.radius, cfg.nsample)):
        _, nx = pn2_ops.furthest_point_sample_gather(xyz, m)
        srt = c.sort_points_x(xyz, 256)
        for r, ns in zip(radii, nss):
            tot = torch.zeros(1, dtype=torch.int32, device="cuda")
            ts = []
            for it in range(12):
                tot.zero_()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                out = c.ball_query_pairs(r, ns, xyz, nx, srt, tot)
                b.record(); torch.cuda.synchronize()
                if it >= 2:
                    ts.append(a.elapsed_time(b) * 1e3)
            line += "  L%d r=%.1f %.1f us" % (lvl + 1, r, float(np.median(ts)))
        xyz = nx
    print(line, flush=True)
