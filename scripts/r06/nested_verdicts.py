"""How often does the verified-prefix sampling of level 2 (4096 -> 1024) fall back to the literal restatement on the headline scenes?
Per scene: ws3d_furthest_point_sampling_nested alone (b = 1), timed: a verified scene costs the three short launches, a fallback adds
~1023 dependent steps (~70 us).  hdl64 scenes of the bench's 20 slots."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from ws3d_amd import compat as c, synth

def t_us(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / it * 1e3

slow = 0; tot = 0; times = []
for slot in range(6):
    pcs = np.stack([synth.cloud("hdl64", 16384, 1000 * 3 + 100000 * slot + s)[:, :3] for s in range(8)])
    xyz = torch.from_numpy(pcs).cuda()
    idx1 = torch.empty((8, 4096), dtype=torch.int32, device="cuda"); x1 = torch.empty((8, 4096, 3), device="cuda")
    c.furthest_point_sampling_gather(8, 16384, 4096, xyz, None, idx1, x1)
    for b in range(8):
        xb = x1[b:b + 1].contiguous()
        i2 = torch.empty((1, 1024), dtype=torch.int32, device="cuda"); n2 = torch.empty((1, 1024, 3), device="cuda")
        us = t_us(lambda: c.furthest_point_sampling_nested(1, 4096, 1024, xb, i2, n2))
        ar = bool(torch.equal(i2[0].cpu(), torch.arange(1024, dtype=torch.int32)))
        times.append(us); tot += 1; slow += us > 120
        print("slot %d scene %d  %.1f us  arange %s" % (slot, b, us, ar))
print("scenes %d, slow (fallback) %d, median %.1f us" % (tot, slow, float(np.median(times))))
