#!/bin/bash
# what bounds sa_mlp3_compact_mfma_kernel (level 1, both scales, batch 8 of hdl64): the kernel as it is, without its atomic epilogue
# (-DSA1_ABL=1), without its gather (=2), without both (=3: the register-chained matrix work alone)
cd "$(dirname "$0")/../.."
OBJ=ws3d_amd/csrc/build
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fvisibility=hidden -Iinclude -Iws3d_amd/csrc"
for abl in 0 1 2 3 ${SA1_EXTRA:-}; do
  hipcc $FLAGS -DSA1_ABL=$abl -c ws3d_amd/csrc/sa_mlp.hip -o /tmp/sa1_abl.o 2>/dev/null || { echo "compile failed ($abl)"; continue; }
  hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libws3d_sa1abl.so $(ls $OBJ/*.o | grep -v "/sa_mlp") /tmp/sa1_abl.o
  WS3D_HIP_LIB=/tmp/libws3d_sa1abl.so SA1_ABL=$abl python - <<'PY'
import os, sys; sys.path.insert(0, ".")
import numpy as np, torch
from ws3d_amd import compat as c, synth
B, N, M = 8, 16384, 4096
pc = np.stack([synth.cloud("hdl64", N, 2000 + s) for s in range(B)])
xyz = torch.from_numpy(pc[:, :, :3].copy()).cuda(); feat = torch.from_numpy(pc[:, :, 3].copy()).cuda()
idx = torch.empty((B, M), dtype=torch.int32, device="cuda"); nx = torch.empty((B, M, 3), device="cuda")
c.furthest_point_sampling_gather(B, N, M, xyz, None, idx, nx)
srt = c.sort_points_x(xyz)
g = torch.Generator(device="cuda").manual_seed(1)
line = "SA1_ABL=%s" % os.environ["SA1_ABL"]
for r, ns, widths in ((0.1, 16, (16, 16, 32)), (0.5, 32, (32, 32, 64))):
    nbr = c.ball_query_lists(r, ns, xyz, nx, srt)
    pairs = c.compact_pairs(nbr)
    T = int(pairs[2].item())
    dims = (4,) + widths
    layers = [(torch.randn((dims[i], dims[i + 1]), device="cuda", generator=g) * 0.3, torch.randn(dims[i + 1], device="cuda", generator=g) * 0.1, True) for i in range(3)]
    out = torch.zeros((B * M, widths[2]), device="cuda")
    for _ in range(5):
        assert c.sa_mlp3_pool_compact(xyz, nx, feat, pairs, layers, out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(50):
        c.sa_mlp3_pool_compact(xyz, nx, feat, pairs, layers, out)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 50
    macs = T * (4 * widths[0] + widths[0] * widths[1] + widths[1] * widths[2])
    line += "   (%d,%d,%d) %d rows: %.1f us = %.1f TFLOP/s" % (widths + (T, us, 2 * macs / us / 1e6))
print(line, flush=True)
PY
done
