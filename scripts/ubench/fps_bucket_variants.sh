#!/bin/bash
# A/B of compile-time variants of the pruned FPS kernel: alternative libraries built on the box, timed with scripts/ab_fps.py
cd "$(dirname "$0")/../.."
OBJ=ws3d_amd/csrc/build
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fvisibility=hidden"
i=0
for v in "$@"; do
  i=$((i+1))
  hipcc $FLAGS $v -c ws3d_amd/csrc/fps_bucket.hip -o /tmp/fb_v$i.o 2>/dev/null
  hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libws3d_fbv$i.so $(ls $OBJ/*.o | grep -v fps_bucket) /tmp/fb_v$i.o
  WS3D_HIP_LIB=/tmp/libws3d_fbv$i.so WS3D_FPS_BUCKET=1 python scripts/ab_fps.py "[$v]" 8x16384x4096 2>/dev/null
done
