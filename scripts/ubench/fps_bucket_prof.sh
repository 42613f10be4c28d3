#!/bin/bash
# per-segment clocks of the pruned FPS kernel (fps_bucket.hip built with -DFB_PROF; the counters come back through `temp`)
cd "$(dirname "$0")/../.."
OBJ=ws3d_amd/csrc/build
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fvisibility=hidden"
hipcc $FLAGS -DFB_PROF -c ws3d_amd/csrc/fps_bucket.hip -o /tmp/fb_prof.o 2>/dev/null
hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libws3d_fbprof.so $(ls $OBJ/*.o | grep -v fps_bucket) /tmp/fb_prof.o
WS3D_HIP_LIB=/tmp/libws3d_fbprof.so WS3D_FPS_BUCKET=1 python - <<'PY'
import sys; sys.path.insert(0, ".")
import numpy as np, torch
from ws3d_amd import compat, synth
B, N, M = 8, 16384, 4096
xyz = torch.from_numpy(np.stack([synth.lidar_cloud(N, 100 + s)[:, :3] for s in range(B)])).cuda()
idx = torch.empty((B, M), dtype=torch.int32, device="cuda"); nx = torch.empty((B, M, 3), device="cuda")
temp = torch.full((B, N), 1e10, device="cuda")
compat.furthest_point_sampling_gather(B, N, M, xyz, temp, idx, nx)
torch.cuda.synchronize()
t = temp[0, :64].cpu().numpy().reshape(8, 8)
win = temp[0, 128:256].cpu().numpy().reshape(16, 8).sum(0)
edges = [1, 32, 64, 128, 256, 512, 1024, 2048, M]
print("buckets updated per step, all 16 waves together, by step window: " + ", ".join("[%d,%d): %.1f" % (edges[i], edges[i + 1], win[i] / max(edges[i + 1] - edges[i], 1)) for i in range(8)))
names = ["bbox test", "updates", "pick", "read+reduce", "active buckets (sum)", "steps with any", "publish", "barrier wait"]
for w in range(8):  # (the first 8 waves)
    print("wave %d: " % w + "  ".join("%s %.0f" % (names[k], t[w, k] / (M - 1)) for k in (0, 1, 2, 6, 7, 3)) + "  | active/step %.2f, steps with any %.2f" % (t[w, 4] / (M - 1), t[w, 5] / (M - 1)))
PY
