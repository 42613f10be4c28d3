"""does the 3-NN search (one lane per query, known set in LDS) run faster when the QUERIES arrive in cell order instead of the order the
network holds them in (scan order at level 0, sampling order -- consecutive points far apart -- at the levels below)?  The queries are
permuted on the host side of the call (the kernel is unchanged): per FP module of a batch of 8 hdl64 scenes.
    python scripts/ubench/three_nn_query_order.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from ws3d_amd import compat as c, synth, stage1, pn2_ops

cfg = stage1.DEFAULT_CFG
lib = c._lib.load()


def binned_order(srt, B, n):
    stride = lib.ws3d_sorted_points_bytes(B, n) // B
    v = srt.view(B, stride)[:, :n * 16].contiguous().view(torch.int32).view(B, n, 4)
    return v[:, :, 3].long()                                  # original index of the point at each binned position


def timeit(fn, reps=30):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps


for kind in ("hdl64", "lidar"):
    B = 8
    xyz = [torch.from_numpy(np.stack([synth.cloud(kind, 16384, 2000 + s)[:, :3] for s in range(B)])).cuda()]
    for m in cfg.npoints:
        xyz.append(pn2_ops.furthest_point_sample_gather(xyz[-1], m)[1])
    line = kind + ":"
    for lvl in range(4):
        unknown, known = xyz[lvl], xyz[lvl + 1]
        n = unknown.size(1)
        sk = c.sort_points_xz(known)
        su = c.sort_points_xz(unknown, min_n=1)
        if su is None:
            line += "  L%d n=%d: no binned copy" % (lvl, n)
            continue
        perm = binned_order(su, B, n)
        uq = torch.gather(unknown, 1, perm.unsqueeze(-1).expand(B, n, 3)).contiguous()
        i0, w0 = c.three_nn_with_weights(unknown, known, sk)
        i1, w1 = c.three_nn_with_weights(uq, known, sk)
        same = torch.equal(torch.gather(i0.long(), 1, perm.unsqueeze(-1).expand(B, n, 3)), i1.long())
        t0 = timeit(lambda: c.three_nn_with_weights(unknown, known, sk))
        t1 = timeit(lambda: c.three_nn_with_weights(uq, known, sk))
        line += "  L%d n=%d m=%d: as held %.1f us, in cell order %.1f us%s" % (lvl, n, known.size(1), t0, t1, "" if same else " (RESULTS DIFFER)")
    print(line, flush=True)
