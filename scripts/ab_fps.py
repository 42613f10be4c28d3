"""A/B timing of furthest_point_sample shapes (env WS3D_FPS_BUCKET / WS3D_FPS_ROUNDS / WS3D_FPS_PAIR read by the library: INTEGRATION.md 4.1).
    python scripts/ab_fps.py [tag]  ->  one line per shape: ms per launch (median of 7), us per step"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ws3d_amd import compat, synth

SHAPES = [(8, 16384, 4096), (256, 16384, 4096), (512, 16384, 4096), (8, 4096, 1024), (8, 1024, 256), (8, 256, 64),
          (64, 4096, 1024), (800, 512, 128), (800, 128, 32), (256, 8192, 2048)]
tag = sys.argv[1] if len(sys.argv) > 1 else ""
shapes = SHAPES if len(sys.argv) <= 2 else [tuple(int(v) for v in a.split("x")) for a in sys.argv[2:]]
for B, N, M in shapes:
    base = np.stack([synth.lidar_cloud(N, 100 + s)[:, :3] for s in range(min(B, 16))])
    xyz = torch.from_numpy(np.ascontiguousarray(np.tile(base, (-(-B // base.shape[0]), 1, 1))[:B])).cuda()
    idx = torch.empty((B, M), dtype=torch.int32, device="cuda")
    nx = torch.empty((B, M, 3), device="cuda")
    temp = torch.empty((B, N), device="cuda") if N > 16384 else None    # the round-1 streaming kernel needs it
    def run():
        if temp is not None:
            temp.fill_(1e10)
        compat.furthest_point_sampling_gather(B, N, M, xyz, temp, idx, nx)
    for _ in range(2):
        run()
    ts = []
    for _ in range(7):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); run(); b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ms = float(np.median(ts))
    print("%-14s B=%4d N=%6d M=%5d  %8.3f ms  %6.3f us/step  checksum %d" % (tag, B, N, M, ms, ms * 1e3 / max(M - 1, 1), int(idx.sum().item())), flush=True)
