import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from ws3d_amd import compat as c, synth
def dev(a): return torch.from_numpy(np.ascontiguousarray(a)).cuda()
B, N, M, C, O1 = 2, 4096, 1024, 96, 64
scales = ((16, 0.5, 64, 128), (32, 1.0, 96, 128))
rng = np.random.default_rng(23)
pc = synth.make_batch("lidar", B, 16384, 71)[:, :N, :3].copy()
xyz = dev(pc)
feats = dev(rng.standard_normal((B, N, C)).astype(np.float32))
idx = torch.empty((B, M), dtype=torch.int32, device="cuda"); new_xyz = torch.empty((B, M, 3), device="cuda")
c.furthest_point_sampling_gather(B, N, M, xyz, None, idx, new_xyz)
srt = c.sort_points_x(xyz)
w1s = [dev((rng.standard_normal((C + 3, O1)) / np.sqrt(C)).astype(np.float32)) for _ in scales]
pmat = feats.view(B * N, C) @ torch.cat([w[:C] for w in w1s], dim=1)
args, col = [], 0
for si, (ns, r, O2, O3) in enumerate(scales):
    nbr = torch.zeros((B, M, ns), dtype=torch.int32, device="cuda")
    c.ball_query_wrapper(B, N, M, r, ns, new_xyz, xyz, nbr, srt)
    args.append({"pmat": pmat, "col0": si * O1, "o1": O1, "xyz": xyz, "new_xyz": new_xyz, "pairs": c.compact_pairs(nbr), "w1x": w1s[si][C:].contiguous(),
                 "b1": dev(rng.standard_normal(O1).astype(np.float32)), "relu1": True,
                 "w2t": dev((rng.standard_normal((O1, O2)) / np.sqrt(O1)).astype(np.float32)), "b2": dev(rng.standard_normal(O2).astype(np.float32)), "relu2": True,
                 "w3t": dev((rng.standard_normal((O2, O3)) / np.sqrt(O2)).astype(np.float32)), "b3": dev(rng.standard_normal(O3).astype(np.float32)),
                 "col_offset": col})
    col += O3
want = torch.zeros((B * M, 256), device="cuda")
for a in args:
    a["out2d"] = want
assert c.compact_mlp_pair(3, args)
print("totals", [int(a["pairs"][2].item()) for a in args], "want nonzero", int((want != 0).sum()))
for trial in range(3):
    out = torch.zeros_like(want)
    for a in args:
        a["out2d"] = out
    t = torch.zeros(c.chain_ticket_ints(), dtype=torch.int32, device="cuda")
    ok = c.chain_mlp3(args, t)
    torch.cuda.synchronize()
    print("trial", trial, "took", ok, "nonzero", int((out != 0).sum()), "equal", bool(torch.equal(out, want)), "ticket", t.view(32, -1)[:, 0].tolist()[:8], "sum", int(t.sum()),
          "scale0 equal", bool(torch.equal(out[:, :128], want[:, :128])), "scale1 equal", bool(torch.equal(out[:, 128:], want[:, 128:])))
