"""Does the fused query + group launch of the c2 block depend on WHERE its buffers lie?  (two bench processes of one box measured 0.96 and
1.05 ms.)  The workload is rebuilt behind paddings of different sizes; per placement: the launch time and the buffers' addresses.
    python scripts/r06/c2_placement.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench_lib
from ws3d_amd import compat as c

B = 512
base = bench_lib.C2(B, 0, "hdl64")
xyz, feat = base.xyz, base.feat
idx = torch.empty((B, 4096), dtype=torch.int32, device="cuda"); new_xyz = torch.empty((B, 4096, 3), device="cuda")
c.furthest_point_sampling_gather(B, 16384, 4096, xyz, None, idx, new_xyz)
del base
for pad_mb in (0, 2, 4, 6, 10, 16, 34, 70, 130, 258, 514, 1026):
    torch.cuda.empty_cache()
    pad = torch.empty(pad_mb << 20, dtype=torch.uint8, device="cuda") if pad_mb else None
    nbr = torch.empty((B, 4096, 64), dtype=torch.int32, device="cuda")
    out = torch.empty((B, 4, 4096, 64), device="cuda")
    ts = []
    for it in range(9):
        a, b, e = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        a.record(); s = c.sort_points_x(xyz); b.record()
        c.query_and_group(B, 16384, 4096, 1, 0.1, 64, True, xyz, new_xyz, feat, nbr, out, s); e.record()
        torch.cuda.synchronize()
        if it >= 2:
            ts.append((a.elapsed_time(b), b.elapsed_time(e)))
    t = np.median(np.array(ts), axis=0)
    print("pad %5d MB  binning %.3f  search + emit %.3f ms   out %#x  lists %#x  binned %#x   (mod 1 GiB: out %4d MB, lists %4d MB)" % (
        pad_mb, t[0], t[1], out.data_ptr(), nbr.data_ptr(), s.data_ptr(), (out.data_ptr() >> 20) & 1023, (nbr.data_ptr() >> 20) & 1023), flush=True)
    del nbr, out, s, pad
