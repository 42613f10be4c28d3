#!/bin/bash
# fps_rounds_kernel built with FR_KMAX = 1 .. 4 (samples certified per round at most): time per launch and rounds -> the fixed cost of a
# round and the cost of each further sample (unprofiled timings; the round counts come from the simulation / the FR_PROF build)
cd "$(dirname "$0")/../.."
OBJ=ws3d_amd/csrc/build
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fvisibility=hidden"
for k in ${KS:-1 2 3 4}; do
  hipcc $FLAGS -DFR_KMAX=$k -c ws3d_amd/csrc/fps_bucket.hip -o /tmp/fr_k$k.o 2>/dev/null
  hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libws3d_frk$k.so $(ls $OBJ/*.o | grep -v fps_bucket) /tmp/fr_k$k.o
  WS3D_HIP_LIB=/tmp/libws3d_frk$k.so python scripts/ab_fps.py kmax=$k ${SHAPES:-8x16384x4096} 2>&1 | grep -v amdgpu
done
