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
                          (256, 16384, 4096), (512, 16384, 4096), (1024, 16384, 4096), (8, 256, 64), (800, 512, 128)]:
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


def s2_case(B):
    """Stage-2 (RCNN) set-abstraction shapes: 512-point RoI clouds (lib/config.py:122-129)"""
    pts = synth.roi_clouds(B, 512, 5)
    xyz = torch.from_numpy(pts).cuda()
    feat = torch.randn((B, 128, 512), device="cuda")
    idx1 = torch.empty((B, 128), dtype=torch.int32, device="cuda"); new1 = torch.empty((B, 128, 3), device="cuda")
    idx2 = torch.empty((B, 32), dtype=torch.int32, device="cuda"); new2 = torch.empty((B, 32, 3), device="cuda")
    nbr1 = torch.empty((B, 128, 64), dtype=torch.int32, device="cuda"); out1 = torch.empty((B, 131, 128, 64), device="cuda")
    nbr2 = torch.empty((B, 32, 64), dtype=torch.int32, device="cuda"); out2 = torch.empty((B, 131, 32, 64), device="cuda")
    feat2 = torch.randn((B, 128, 128), device="cuda")
    t = {}
    t["fps1"] = timeit(lambda: c.furthest_point_sampling_gather(B, 512, 128, xyz, None, idx1, new1))[0]
    t["qg1"] = timeit(lambda: c.query_and_group(B, 512, 128, 128, 0.2, 64, True, xyz, new1, feat, nbr1, out1, None))[0]
    t["fps2"] = timeit(lambda: c.furthest_point_sampling_gather(B, 128, 32, new1, None, idx2, new2))[0]
    t["qg2"] = timeit(lambda: c.query_and_group(B, 128, 32, 128, 0.4, 64, True, new1, new2, feat2, nbr2, out2, None))[0]
    gb1, gb2 = out1.numel() * 4 / 1e9, out2.numel() * 4 / 1e9
    print(f"s2 B={B}: " + " ".join(f"{k}={v:.3f}ms" for k, v in t.items()) +
          f"  qg1 {gb1 / t['qg1'] * 1e3:.0f} GB/s  qg2 {gb2 / t['qg2'] * 1e3:.0f} GB/s (output bytes only)")


if __name__ == "__main__" and "--what" in sys.argv and sys.argv[sys.argv.index("--what") + 1] == "s2":
    for B in (64, 800):
        s2_case(B)


def copy_ops_case(B):
    """the standalone copy kernels at Stage-2 / Stage-1 shapes: GB/s of bytes written"""
    xyz = torch.from_numpy(synth.roi_clouds(B, 512, 5)).cuda()
    feat = torch.randn((B, 128, 512), device="cuda")
    idx = torch.randint(0, 512, (B, 128, 64), dtype=torch.int32, device="cuda")
    out = torch.empty((B, 128, 128, 64), device="cuda")
    t = timeit(lambda: c.group_points_wrapper(B, 128, 512, 128, 64, feat, idx, out))[0]
    print(f"group_points B={B} C=128 N=512 M=128 ns=64: {t:.3f} ms  {out.numel() * 4 / t / 1e6:.0f} GB/s written")
    gi = torch.randint(0, 512, (B, 128), dtype=torch.int32, device="cuda")
    go = torch.empty((B, 128, 128), device="cuda")
    t = timeit(lambda: c.gather_points_wrapper(B, 128, 512, 128, feat, gi, go))[0]
    print(f"gather_points B={B} C=128 N=512 M=128: {t:.3f} ms  {go.numel() * 4 / t / 1e6:.0f} GB/s written")
    B2 = max(B // 100, 1)
    known = torch.randn((B2, 256, 4096), device="cuda")
    ii = torch.randint(0, 4096, (B2, 16384, 3), dtype=torch.int32, device="cuda")
    ww = torch.rand((B2, 16384, 3), device="cuda")
    oo = torch.empty((B2, 256, 16384), device="cuda")
    t = timeit(lambda: c.three_interpolate_wrapper(B2, 256, 4096, 16384, known, ii, ww, oo))[0]
    print(f"three_interpolate B={B2} C=256 M=4096 N=16384: {t:.3f} ms  {oo.numel() * 4 / t / 1e6:.0f} GB/s written")


if __name__ == "__main__" and "--what" in sys.argv and sys.argv[sys.argv.index("--what") + 1] == "copy":
    for B in (64, 800):
        copy_ops_case(B)


def nn_case(B, N, M):
    pc = synth.make_batch("lidar", B, N, 2)[:, :, :3].copy()
    unk = torch.from_numpy(pc).cuda()
    idx = torch.empty((B, M), dtype=torch.int32, device="cuda"); kn = torch.empty((B, M, 3), device="cuda")
    c.furthest_point_sampling_gather(B, N, M, unk, None, idx, kn)
    d2 = torch.empty((B, N, 3), device="cuda"); i2 = torch.empty((B, N, 3), dtype=torch.int32, device="cuda")
    t_b = timeit(lambda: c.three_nn_wrapper(B, N, M, unk, kn, d2, i2))[0]
    srt = c.sort_points_x(kn, min_n=64)
    if srt is None:
        print(f"three_nn B={B} N={N} M={M}: brute {t_b:.3f} ms")
        return
    t_sort = timeit(lambda: c.sort_points_x(kn, min_n=64))[0]
    t_s = timeit(lambda: c.three_nn_wrapper(B, N, M, unk, kn, d2, i2, srt))[0]
    ref = i2.clone()
    grid = c.sort_points_xz(kn, min_n=1)
    t_gsort = timeit(lambda: c.sort_points_xz(kn, min_n=1))[0]
    t_g = timeit(lambda: c.three_nn_wrapper(B, N, M, unk, kn, d2, i2, grid))[0]
    print(f"three_nn B={B} N={N} M={M}: brute {t_b:.3f} ms, x-binned {t_s:.3f} ms + sort {t_sort:.3f} ms, "
          f"xz-grid {t_g:.3f} ms + sort {t_gsort:.3f} ms (same idx: {bool(torch.equal(ref, i2))})")


if __name__ == "__main__" and "--what" in sys.argv and sys.argv[sys.argv.index("--what") + 1] == "nn":
    for (B, N, M) in [(8, 16384, 4096), (8, 4096, 1024), (8, 1024, 256), (8, 256, 64), (64, 16384, 4096)]:
        nn_case(B, N, M)


def grad_case(B, C, N, M, ns, radius=None):
    """backward of group_points at the training shapes; idx from a real ball query on scan-like
    scenes when radius is given (padding repeats the first neighbour: very uneven segments)"""
    if radius is None:
        idx = torch.randint(0, N, (B, M, ns), dtype=torch.int32, device="cuda")
    else:
        from ws3d_amd import synth, pn2_ops
        pts = torch.from_numpy(np.stack([synth.velodyne_scan(16384, seed=i)[:, :3] for i in range(B)])).cuda()
        lvl = pts
        while lvl.size(1) > N:
            _, lvl = pn2_ops.furthest_point_sample_gather(lvl, lvl.size(1) // 4)
        _, ctr = pn2_ops.furthest_point_sample_gather(lvl, M)
        idx = pn2_ops.ball_query(radius, ns, lvl, ctr)
    g = torch.randn((B, C, M, ns), device="cuda")
    out = torch.zeros((B, C, N), device="cuda")
    ta = timeit(lambda: (out.zero_(), c.group_points_grad_wrapper(B, C, N, M, ns, g, idx, out)))[0]
    ref = out.clone()
    td = timeit(lambda: c.group_points_grad_det(B, C, N, M, ns, g, idx, out))[0]
    print(f"group_points_grad B={B} C={C} N={N} M={M} ns={ns} r={radius}: atomic {ta:.3f} ms, deterministic {td:.3f} ms"
          f"  (max |diff| {float((ref - out).abs().max()):.2e})")


def interp_grad_case(B, C, n, m):
    from ws3d_amd import pn2_ops
    pts = torch.from_numpy(np.stack([synth.velodyne_scan(16384, seed=i)[:, :3] for i in range(B)])).cuda()
    unk = pts
    while unk.size(1) > n:
        _, unk = pn2_ops.furthest_point_sample_gather(unk, unk.size(1) // 4)
    _, kn = pn2_ops.furthest_point_sample_gather(unk, m)
    dist, idx = pn2_ops.three_nn(unk, kn)
    w = 1.0 / (dist + 1e-8); w = (w / w.sum(2, keepdim=True)).contiguous()
    g = torch.randn((B, C, n), device="cuda")
    out = torch.zeros((B, C, m), device="cuda")
    ta = timeit(lambda: (out.zero_(), c.three_interpolate_grad_wrapper(B, C, n, m, g, idx, w, out)))[0]
    td = timeit(lambda: c.three_interpolate_grad_det(B, C, n, m, g, idx, w, out))[0]
    print(f"three_interpolate_grad B={B} C={C} n={n} m={m}: atomic {ta:.3f} ms, deterministic {td:.3f} ms")


if __name__ == "__main__" and "--what" in sys.argv and sys.argv[sys.argv.index("--what") + 1] == "grad":
    for shp in [(8, 256, 16384, 4096), (8, 512, 4096, 1024), (8, 512, 1024, 256), (8, 512, 256, 64)]:
        interp_grad_case(*shp)
    for shp in [(8, 96, 4096, 1024, 16, 0.5), (8, 96, 4096, 1024, 32, 1.0), (8, 256, 1024, 256, 16, 1.0),
                (8, 256, 1024, 256, 32, 2.0), (8, 512, 256, 64, 16, 2.0), (8, 512, 256, 64, 32, 4.0),
                (16, 64, 16384, 4096, 32, None), (16, 128, 4096, 1024, 32, None)]:
        grad_case(*shp)
