#!/bin/bash
# per-segment clocks of fps_rounds_kernel (fps_bucket.hip built with -DFR_PROF; the counters come back through `temp`) and the samples per round
cd "$(dirname "$0")/../.."
OBJ=ws3d_amd/csrc/build
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fvisibility=hidden"
hipcc $FLAGS -DFR_PROF ${FR_EXTRA:-} -c ws3d_amd/csrc/fps_bucket.hip -o /tmp/fr_prof.o 2>/dev/null
hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libws3d_frprof.so $(ls $OBJ/*.o | grep -v fps_bucket) /tmp/fr_prof.o
WS3D_HIP_LIB=/tmp/libws3d_frprof.so WS3D_FPS_BUCKET=1 WS3D_FPS_ROUNDS=1 python - <<'PY'
import sys; sys.path.insert(0, ".")
import numpy as np, torch
from ws3d_amd import compat, synth
B, N, M = 8, 16384, 4096
for kind in ("hdl64", "lidar"):
    xyz = torch.from_numpy(np.stack([synth.cloud(kind, N, 100 + s)[:, :3] for s in range(B)])).cuda()
    idx = torch.empty((B, M), dtype=torch.int32, device="cuda"); nx = torch.empty((B, M, 3), device="cuda")
    temp = torch.full((B, N), 1e10, device="cuda")
    compat.furthest_point_sampling_gather(B, N, M, xyz, temp, idx, nx)
    torch.cuda.synchronize()
    t = temp[0, :128].cpu().numpy().reshape(16, 8)
    kh = temp[0, 128:256].cpu().numpy().reshape(16, 8)[0]
    rounds = kh.sum()
    print("%s: %d rounds for %d samples = %.2f samples per round; rounds by number of samples (tie-round, 1, 2, 3, 4): %s" % (kind, rounds, M - 1, (M - 1) / rounds, [int(kh[k]) for k in (0, 1, 2, 3, 4)]))
    names = ["box tests", "updates", "re-pick", "publish", "wait A", "certify / idle", "wait B", "read samples"]
    for w in (0, 5, 15):
        print("  wave %2d, clk per round: " % w + "  ".join("%s %.0f" % (names[k], t[w, k] / rounds) for k in range(8)) + "  | sum %.0f" % (t[w].sum() / rounds))
PY
