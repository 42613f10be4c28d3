#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r04c7; mkdir -p $OUT; export TMPDIR=/tmp
for d in 12 16 20 22; do
  timeout 600 python bench.py --pipeline-depth $d --no-cpu-baseline --no-side-runs --c2-batch 0 2>/dev/null | tail -1 > $OUT/depth_$d.json
  python -c "
import json; d=json.load(open('$OUT/depth_$d.json')); print('depth', $d, 'value %.0f' % d['value'], 'ms/batch %.4f' % d['ms_per_step'], 'latency %.3f' % d['latency_mode']['ms_per_batch'])"
done
