#!/bin/bash
# scripts/ablate.sh <source.hip> <out.so> [-DFLAG ...]: rebuild ONE translation unit with extra
# defines and link it with the other objects into scratch/<out.so> (select with WS3D_HIP_LIB=...).
set -e
cd "$(dirname "$0")/.."
src=$1; out=$2; shift 2
mkdir -p scratch
python -c "import ws3d_amd.build as b; b.build()"
objs=""
for s in core fps fps_bucket ballquery_group interpolate roipool3d iou3d scatter_det sa_mlp; do
  [ "$s.hip" = "$src" ] || objs="$objs ws3d_amd/csrc/build/$s.o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fvisibility=hidden \
  -Iinclude "$@" -c ws3d_amd/csrc/$src -o scratch/abl_tmp.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scratch/$out scratch/abl_tmp.o $objs
echo scratch/$out
