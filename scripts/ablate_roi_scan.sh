#!/bin/bash
# What the scan of roipool3d_pipe_kernel spends its time on (c5 shape, copy phase skipped): full scan / tests without the list
# appends / loads and loop only -- alternative libraries built on the box.
cd "$(dirname "$0")/.."
OBJ=ws3d_amd/csrc/build
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fvisibility=hidden"
for v in 0 1 2; do
  hipcc $FLAGS -DWS3D_ROI_NO_COPY -DWS3D_ROI_SCAN_ABL=$v -c ws3d_amd/csrc/roipool3d.hip -o /tmp/roi_s$v.o 2>/dev/null
  hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libws3d_s$v.so $(ls $OBJ/*.o | grep -v roipool3d) /tmp/roi_s$v.o
  line=$(WS3D_HIP_LIB=/tmp/libws3d_s$v.so timeout 200 python bench.py --full-line --workload c5 --no-cpu-baseline 2>/dev/null | tail -1)
  echo "scan ablation $v (0 full, 1 no appends, 2 loads + loop only): $(echo "$line" | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=[x for x in d['kernels'] if 'roipool' in x['name']][0]; print('kernel ms', round(k['ms_per_step'],4))")"
done
