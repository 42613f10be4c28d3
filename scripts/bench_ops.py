"""Micro-timings of individual ops on the GPU box (HIP events, torch current stream)."""
import argparse, os, sys, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ws3d_amd import compat as c, synth


def timeit(fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); b.synchronize()
        ts.append(a.elapsed_time(b))
    return float(np.median(ts)), float(np.min(ts))


def fps_case(B, N, M):
    pc = synth.make_batch("lidar", min(B, 8), N, 2)[:, :, :3]
    pc = np.ascontiguousarray(np.tile(pc, ((B + pc.shape[0] - 1) // pc.shape[0], 1, 1))[:B])
    xyz = torch.from_numpy(pc).cuda()
    idx = torch.empty((B, M), dtype=torch.int32, device="cuda")
    new = torch.empty((B, M, 3), device="cuda")
    temp = torch.full((B, N), 1e10, device="cuda") if N > 16384 else None
    med, mn = timeit(lambda: c.furthest_point_sampling_gather(B, N, M, xyz, temp, idx, new))
    print(f"fps B={B} N={N} M={M}: {med:.3f} ms (min {mn:.3f})  {med*1e3/(M-1):.3f} us/step")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--what", default="fps")
    a = ap.parse_args()
    if a.what == "fps":
        for (B, N, M) in [(8, 16384, 4096), (8, 8192, 4096), (8, 4096, 1024), (256, 4096, 1024), (8, 1024, 256),
                          (256, 16384, 4096), (8, 256, 64), (800, 512, 128)]:
            fps_case(B, N, M)


def bq_case(B, N, M, r, ns, C, use_sorted):
    pc = synth.make_batch("lidar", min(B, 8), N, 3)
    pc = np.ascontiguousarray(np.tile(pc, ((B + pc.shape[0] - 1) // pc.shape[0], 1, 1))[:B])
    xyz = torch.from_numpy(pc[:, :, :3].copy()).cuda()
    feat = torch.randn((B, C, N), device="cuda")
    idx = torch.empty((B, M), dtype=torch.int32, device="cuda")
    new = torch.empty((B, M, 3), device="cuda")
    c.furthest_point_sampling_gather(B, N, M, xyz, None, idx, new)
    nbr = torch.empty((B, M, ns), dtype=torch.int32, device="cuda")
    out = torch.empty((B, 3 + C, M, ns), device="cuda")
    if use_sorted:
        s_med, _ = timeit(lambda: c.sort_points_x(xyz))
        srt = c.sort_points_x(xyz)
    else:
        s_med, srt = 0.0, None
    med, mn = timeit(lambda: c.query_and_group(B, N, M, C, r, ns, True, xyz, new, feat, nbr, out, srt))
    print(f"qg B={B} N={N} M={M} r={r} ns={ns} C={C} sorted={use_sorted}: query {med:.3f} ms, sort {s_med:.3f} ms")


if __name__ == "__main__" and "--what" in sys.argv and sys.argv[sys.argv.index("--what") + 1] == "bq":
    for B in (8, 256):
        for (N, M, r, ns, C) in [(16384, 4096, 0.1, 16, 1), (16384, 4096, 0.5, 32, 1), (4096, 1024, 0.5, 16, 96),
                                 (4096, 1024, 1.0, 32, 96), (16384, 4096, 0.1, 64, 1)]:
            if B == 256 and C == 96:
                continue
            for srt in (False, True):
                bq_case(B, N, M, r, ns, C, srt)
