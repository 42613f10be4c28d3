#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r04c9; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_gpu_fullsize.py -m gpu -x -q -k "fps or FPS or sampling or golden" > $OUT/pytest_fps.log 2>&1; tail -3 $OUT/pytest_fps.log
timeout 900 python scripts/fuzz_parity.py 400 > $OUT/fuzz_parity.txt 2>&1; tail -3 $OUT/fuzz_parity.txt
timeout 900 python bench.py 2>$OUT/bench_default.err | tail -1 > $OUT/bench_default.json
timeout 900 python bench.py --kind lidar 2>/dev/null | tail -1 > $OUT/bench_default_lidar.json
python -c "
import json
for f in ('bench_default','bench_default_lidar'):
    d=json.load(open('$OUT/'+f+'.json')); r=d['roofline']; print(f,'value %.0f' % d['value'], 'latency %.3f' % d['latency_mode']['ms_per_batch'], 'c2 %.0f' % d['c2']['scenes_per_s_per_gpu'], 'roofline frac', r['frac'], 'traffic', r['traffic'], r['frac_is'])"
