"""the fused query + group launch of the c2 block (512 scenes, 16384 -> 4096 centres, r = 0.1, 64 samples, 3 + 1 channels) timed alone,
binning and search + emit separately, on both generators.    python scripts/r06/bq_time.py [label]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from ws3d_amd import compat, synth
label = sys.argv[1] if len(sys.argv) > 1 else "default"
B, N, M, r, ns, C = 512, 16384, 4096, 0.1, 64, 1
for kind in ("hdl64", "lidar"):
    base = np.stack([synth.cloud(kind, N, 2000 + s) for s in range(16)])
    pc = np.ascontiguousarray(np.tile(base, (B // 16, 1, 1)))
    xyz = torch.from_numpy(pc[:, :, :3].copy()).cuda(); feat = torch.from_numpy(np.ascontiguousarray(pc[:, :, 3:4].transpose(0, 2, 1))).cuda()
    idx = torch.empty((B, M), dtype=torch.int32, device="cuda"); new_xyz = torch.empty((B, M, 3), device="cuda")
    compat.furthest_point_sampling_gather(B, N, M, xyz, None, idx, new_xyz)
    nbr = torch.empty((B, M, ns), dtype=torch.int32, device="cuda"); out = torch.empty((B, 3 + C, M, ns), device="cuda")
    ts = []
    for it in range(11):
        a, b, c = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        a.record(); s = compat.sort_points_x(xyz, grid=True); b.record()
        compat.query_and_group(B, N, M, C, r, ns, True, xyz, new_xyz, feat, nbr, out, s); c.record()
        torch.cuda.synchronize()
        if it >= 3:
            ts.append((a.elapsed_time(b), b.elapsed_time(c)))
    t = np.median(np.array(ts), axis=0)
    gb = (M * ns * 4 + (3 + C) * N * 4 + (3 + C) * M * ns * 4 + (N + M) * 12) * B / 1e9
    print("%-22s %-6s binning %.3f ms  search + emit %.3f ms  -> %.3f of 8 TB/s incl. binning   checksum %d %.6e" % (
        label, kind, t[0], t[1], gb / (t.sum() * 1e-3) / 8e3, int(nbr.long().sum().item()), float(out.double().sum().item())), flush=True)
