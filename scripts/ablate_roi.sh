#!/bin/bash
# Ablation of roipool3d at the c5 shape: full / no scan / no copy, with the copy phase direct (WS3D_ROI_STAGE=0) or staged
# through LDS (16 / 32 rows) -- alternative libraries built on the box, kernel time from bench.py's HIP events.
cd "$(dirname "$0")/.."
OBJ=ws3d_amd/csrc/build
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fvisibility=hidden"
for v in NO_SCAN NO_COPY; do
  hipcc $FLAGS -DWS3D_ROI_$v -c ws3d_amd/csrc/roipool3d.hip -o /tmp/roi_$v.o 2>/dev/null
  hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libws3d_$v.so $(ls $OBJ/*.o | grep -v roipool3d) /tmp/roi_$v.o
done
for v in FULL NO_SCAN NO_COPY; do
  for st in 16; do for pp in 0 2; do
    lib=""; [ $v != FULL ] && lib=/tmp/libws3d_$v.so
    line=$(WS3D_HIP_LIB=$lib WS3D_ROI_PIPE=$pp WS3D_ROI_STAGE=$st timeout 200 python bench.py --full-line --workload c5 --no-cpu-baseline 2>/dev/null | tail -1)
    echo "$v stage=$st pipe=$pp $(echo "$line" | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=[x for x in d['kernels'] if 'roipool' in x['name']][0]; print('kernel ms', round(k['ms_per_step'],4), 'step ms', round(d['ms_per_step'],4))")"
  done; done
done
