#!/bin/bash
# Critical-path anatomy of a round of fps_rounds_kernel WITHOUT clock hooks: builds (-DFR_DUP=k) that execute segment k of every
# round twice (each segment is idempotent, results stay bit-identical: same checksum).  (time - plain time) / rounds = what the
# segment adds to the round.  Segments: 0 box tests, 1 + 3 bucket updates (slots 0-7, 8-15), 2 re-pick, 4 barrier A, 5 certification, 6 barrier B,
# 7 reading the round's samples; 15 = none (the plain round).
cd "$(dirname "$0")/../.."
OBJ=ws3d_amd/csrc/build
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fvisibility=hidden"
for seg in 15 0 1 3 2 4 5 6 7; do
  hipcc $FLAGS -DFR_DUP=$seg -c ws3d_amd/csrc/fps_bucket.hip -o /tmp/fr_dup$seg.o 2>/dev/null &
done
wait
for seg in 15 0 1 3 2 4 5 6 7 15; do
  hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libws3d_frdup.so $(ls $OBJ/*.o | grep -v fps_bucket) /tmp/fr_dup$seg.o
  WS3D_HIP_LIB=/tmp/libws3d_frdup.so python scripts/ab_fps.py dup=$seg ${1:-8x16384x4096} 2>&1 | grep -v amdgpu
done
