#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r04c13; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python bench.py 2>$OUT/bench_default.err | tail -1 > $OUT/bench_default.json
timeout 900 python bench.py --kind lidar 2>/dev/null | tail -1 > $OUT/bench_default_lidar.json
timeout 900 python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $OUT/bench_default_steps20.json
python -c "
import json
for f in ('bench_default','bench_default_lidar','bench_default_steps20'):
    d=json.load(open('$OUT/'+f+'.json')); r=d['roofline']; print(f,'value %.0f' % d['value'], 'latency %.3f' % d['latency_mode']['ms_per_batch'], 'c2 %.0f' % d['c2']['scenes_per_s_per_gpu'], 'roofline frac', r['frac'], 'traffic', r['traffic'])"
