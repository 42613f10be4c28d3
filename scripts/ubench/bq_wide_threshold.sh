#!/bin/bash
# ball_query_grid_coop_kernel: 16 waves x 4 centres per tile at EVERY launch size (-DBQC_WIDE_BELOW=2000000000) against the default (below 2048 tiles):
# the c2 block (512 scenes = 32768 tiles) and the eight searches of a c3 batch
cd "$(dirname "$0")/../.."
OBJ=ws3d_amd/csrc/build
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fvisibility=hidden -Iinclude -Iws3d_amd/csrc"
hipcc $FLAGS ${BQ_DEFS:--DBQC_WIDE_BELOW=2000000000} -c ws3d_amd/csrc/ballquery_group.hip -o /tmp/bq_wide.o 2>/dev/null || { echo "compile failed"; exit 1; }
hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libws3d_bqwide.so $(ls $OBJ/*.o | grep -v "/ballquery_group") /tmp/bq_wide.o
for rep in 1 2; do
for lib in default wide; do
  if [ $lib = wide ]; then export WS3D_HIP_LIB=/tmp/libws3d_bqwide.so; else unset WS3D_HIP_LIB; fi
  for kind in hdl64 lidar; do
  python bench.py --workload c2 --kind $kind --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); k = d['kernels']
print('$lib $kind c2 512 scenes: %.0f scenes/s, sampling %.3f ms, binning + query + group %.3f ms = %.3f of 8 TB/s' % (d['value'], k[0]['ms_per_step'], k[1]['ms_per_step'], k[1]['frac_of_8TBps']))"
  done
done; done
