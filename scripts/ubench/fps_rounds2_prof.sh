#!/bin/bash
# per-segment clocks of fps_rounds2_kernel (fps_bucket.hip built with -DFR2_PROF; the counters come back through `temp`), bucket updates
# per round and wave, and the samples per round
cd "$(dirname "$0")/../.."
OBJ=ws3d_amd/csrc/build
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fvisibility=hidden"
hipcc $FLAGS -DFR2_PROF ${FR_EXTRA:-} -c ws3d_amd/csrc/fps_bucket.hip -o /tmp/fr2_prof.o 2>/dev/null
hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libws3d_fr2prof.so $(ls $OBJ/*.o | grep -v fps_bucket) /tmp/fr2_prof.o
WS3D_HIP_LIB=/tmp/libws3d_fr2prof.so python - <<'PY'
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
    khall = temp[0, 128:128 + 192].cpu().numpy().reshape(16, 12)
    kh = khall[0].copy(); repicks = khall[:, 11].copy(); kh[11] = 0
    rounds = kh.sum()
    print("%s: %d rounds for %d samples = %.2f samples per round; rounds by number of samples (tie-round, 1, 2, ...): %s" % (kind, rounds, M - 1, (M - 1) / rounds, [int(v) for v in kh]))
    print("  re-picks per round: %.1f of 16 waves (busiest wave %.2f)" % (repicks.sum() / rounds, repicks.max() / rounds))
    print("  bucket updates per round: %.1f in all 16 waves, busiest wave %.2f, average wave %.2f" % (t[:, 7].sum() / rounds, t[:, 7].max() / rounds, t[:, 7].mean() / rounds))
    names = ["box tests", "updates", "re-pick + publish", "wait A", "certify / idle", "wait B", "read samples"]
    cc = temp[0, 512:520].cpu().numpy()
    print("  certification (wave 0), clk per round: load candidates %.0f  ranks %.0f  scatter + bound + read back %.0f  condition loop %.0f  count + park %.0f" % tuple(cc[:5] / rounds))
    for w in (0, 5, 15):
        print("  wave %2d, clk per round: " % w + "  ".join("%s %.0f" % (names[k], t[w, k] / rounds) for k in range(7)) + "  | sum %.0f" % (t[w, :7].sum() / rounds))
PY
