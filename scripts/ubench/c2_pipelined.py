"""c2 (FPS 16384 -> 4096 + fused ball query / group, 512 scenes per launch) with the steps of CONSECUTIVE batches on alternating HIP
streams: the sampling kernel issues a quarter of the VALU slots and touches no memory, the query + group kernel is bound by its
stores -- side by side on the same CUs they should cost little more than the sampling alone.
    python scripts/ubench/c2_pipelined.py [batch] [kind]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from ws3d_amd import compat as c, synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
kind = sys.argv[2] if len(sys.argv) > 2 else "hdl64"
N, M, NS, CF, R = 16384, 4096, 64, 1, 0.1
base = np.stack([synth.cloud(kind, N, 2000 + s) for s in range(min(B, 32))])
pc = np.tile(base, (-(-B // base.shape[0]), 1, 1))[:B]
xyz = torch.from_numpy(np.ascontiguousarray(pc[:, :, :3])).cuda()
feat = torch.from_numpy(np.ascontiguousarray(pc[:, :, 3:].transpose(0, 2, 1))).cuda()


def slot():
    return dict(idx=torch.empty((B, M), dtype=torch.int32, device="cuda"), new_xyz=torch.empty((B, M, 3), device="cuda"),
                nbr=torch.empty((B, M, NS), dtype=torch.int32, device="cuda"), grouped=torch.empty((B, 3 + CF, M, NS), device="cuda"),
                stream=torch.cuda.Stream())


def step(s):
    c.furthest_point_sampling_gather(B, N, M, xyz, None, s["idx"], s["new_xyz"])
    c.query_and_group(B, N, M, CF, R, NS, True, xyz, s["new_xyz"], feat, s["nbr"], s["grouped"], c.sort_points_x(xyz))


for depth in (1, 2, 3):
    slots = [slot() for _ in range(depth)]
    for s in slots:
        with torch.cuda.stream(s["stream"]):
            step(s)
    torch.cuda.synchronize()
    K = 12
    t0 = time.perf_counter()
    for k in range(K):
        s = slots[k % depth]
        with torch.cuda.stream(s["stream"]):
            step(s)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    print("depth %d: %.3f ms per step, %.0f scenes/s" % (depth, dt * 1e3, B / dt), flush=True)
    ref = slots[0]["grouped"][:4].clone() if depth == 1 else ref
    assert torch.equal(slots[-1]["grouped"][:4], ref)
    del slots
    torch.cuda.empty_cache()
