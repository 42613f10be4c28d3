#!/bin/bash
# Ablation of the binned ball query + group (c2 shape): full / no search / no emit -- alternative libraries built on the box.
set -e
cd "$(dirname "$0")/.."
OBJ=ws3d_amd/csrc/build
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fvisibility=hidden"
for v in NO_SEARCH NO_EMIT PACKED_ABL; do
  hipcc $FLAGS -DWS3D_BQS_$v -DWS3D_BQG_$v -c ws3d_amd/csrc/ballquery_group.hip -o /tmp/bq_$v.o
  hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libws3d_$v.so $(ls $OBJ/*.o | grep -v ballquery_group) /tmp/bq_$v.o
done
for v in FULL NO_SEARCH NO_EMIT PACKED_ABL; do
  WS3D_ALT_LIB=$([ $v = FULL ] && echo "" || echo /tmp/libws3d_$v.so) python - <<PY
import os, sys
sys.path.insert(0, ".")
import numpy as np, torch
from ws3d_amd import _lib
if os.environ.get("WS3D_ALT_LIB"):
    _lib.LIB_PATH = os.environ["WS3D_ALT_LIB"]
from ws3d_amd import compat, synth
B, N, M, r, ns, C = 512, 16384, 4096, 0.1, 64, 1
base = np.stack([synth.lidar_cloud(N, 200 + s) for s in range(16)])
pc = np.ascontiguousarray(np.tile(base, (B // 16, 1, 1)))
xyz = torch.from_numpy(pc[:, :, :3].copy()).cuda(); feat = torch.randn((B, C, N), device="cuda")
idx = torch.empty((B, M), dtype=torch.int32, device="cuda"); new_xyz = torch.empty((B, M, 3), device="cuda")
compat.furthest_point_sampling_gather(B, N, M, xyz, None, idx, new_xyz)
nbr = torch.empty((B, M, ns), dtype=torch.int32, device="cuda"); out = torch.empty((B, 3 + C, M, ns), device="cuda")
for grid in (True, False):
    s = compat.sort_points_x(xyz, grid=grid)
    ts = []
    for it in range(7):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); compat.query_and_group(B, N, M, C, r, ns, True, xyz, new_xyz, feat, nbr, out, s); b.record()
        torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    print("$v", "grid" if grid else "xslab", "%.3f ms" % np.median(ts), flush=True)
PY
done
