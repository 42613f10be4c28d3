#!/bin/bash
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
O=gpurun_out/r04c25; mkdir -p $O
for rep in 1 2; do
for lib in old new; do
  if [ $lib = old ]; then export WS3D_HIP_LIB=$PWD/ws3d_amd/libws3d_hip_old.so; else unset WS3D_HIP_LIB; fi
  timeout 600 python bench.py --no-side-runs --no-cpu-baseline --c2-batch 0 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1])
print('$lib', round(d['value']), 'scenes/s  latency %.3f ms' % d['latency_mode']['ms_per_batch'])"
done; done
