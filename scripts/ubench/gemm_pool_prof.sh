#!/bin/bash
# Where a wave of the gemm_pool tile loop spends its clocks (gemm_pool.hip built with -DGP_PROF: s_memtime stamps per wave):
# prologue (first loads + first barrier), matrix section (LDS reads + MFMA issue), stage (wait for the next tile's loads + LDS
# writes), barrier (includes the drain of the issued MFMAs), epilogue; and how many waves shared a SIMD.
cd "$(dirname "$0")/../.."
OBJ=ws3d_amd/csrc/build
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fvisibility=hidden -Iinclude -Iws3d_amd/csrc"
hipcc $FLAGS -DGP_PROF -c ws3d_amd/csrc/gemm_pool.hip -o /tmp/gp_prof.o 2>/dev/null || exit 1
hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libws3d_gpprof.so $(ls $OBJ/*.o | grep -v gemm_pool) /tmp/gp_prof.o || exit 1
for tile in ${TILES:-22 21 11}; do
WS3D_HIP_LIB=/tmp/libws3d_gpprof.so WS3D_GP_TILE=$tile WS3D_GP_FORCE_BIG=1 python - $tile <<'PY'
import sys; sys.path.insert(0, ".")
import ctypes as C, numpy as np, torch
from ws3d_amd import compat as c, _lib
tile = int(sys.argv[1]); mb, nb = tile // 10, tile % 10
lib = _lib.load(); rd = lib.ws3d_gp_prof_read; rd.restype = C.c_int; rd.argtypes = [C.c_void_p, C.c_long]
for (rows, ns, k, o) in [(262144, 32, 96, 128), (65536, 32, 196, 256), (16384, 32, 384, 512)]:
    x = torch.randn(rows, k, device="cuda"); wt = torch.randn(k, o, device="cuda") * 0.1; bias = torch.randn(o, device="cuda")
    out = torch.empty(rows // ns, o, device="cuda")
    for _ in range(3): c.gemm_pool(x, wt, bias, True, ns, out, 0)
    torch.cuda.synchronize()
    a = torch.cuda.Event(True); b = torch.cuda.Event(True); a.record(); c.gemm_pool(x, wt, bias, True, ns, out, 0); b.record(); torch.cuda.synchronize()
    nw = min(65536, rows // (64 * mb) * (o // (64 * nb)) * 4)
    buf = np.zeros(nw * 8, dtype=np.int64); assert rd(buf.ctypes.data, nw * 8) == 0
    q = buf.reshape(nw, 8).astype(np.float64)
    t0 = q[:, 0].min(); total = q[:, 5].max() - t0
    kt = (k + 15) // 16
    mfma_clk = kt * 8 * mb * nb * 64
    print(f"tile {tile} rows {rows} k {k} o {o}: event {a.elapsed_time(b) * 1e3:.1f} us, per wave (mean clk): "
          f"resident {np.mean(q[:, 5] - q[:, 0]):.0f} = prologue {np.mean(q[:, 1] - q[:, 0]):.0f} + matrix {np.mean(q[:, 2]):.0f} + stage {np.mean(q[:, 3]):.0f} "
          f"+ barrier {np.mean(q[:, 4]):.0f};  MFMA pipe time of one wave {mfma_clk} clk; k-tiles {kt}")
    clock = (q[:, 5] - q[:, 0]) / (q[:, 7] / 100e6) / 1e9      # shader clocks per real-time second while the wave ran
    print(f"   shader clock while resident: mean {clock.mean():.2f} GHz (min {clock.min():.2f}, max {clock.max():.2f}); "
          f"resident time {np.mean(q[:, 7]) / 100:.1f} us mean, {np.max(q[:, 7]) / 100:.1f} us max")
PY
done
