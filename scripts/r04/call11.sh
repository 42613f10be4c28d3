#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r04c11; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_gpu_fullsize.py -m gpu -x -q -k "nms or NMS or proposals or golden or stage1 or launch_mode" > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
for i in 1 2; do timeout 600 python bench.py --no-cpu-baseline --no-side-runs --c2-batch 0 2>/dev/null | tail -1 > $OUT/bench_$i.json; python -c "
import json; d=json.load(open('$OUT/bench_$i.json')); print('value %.0f' % d['value'], 'latency %.3f' % d['latency_mode']['ms_per_batch'], [ (k['name'][:12], round(k['ms_per_step'],4)) for k in d['kernels'] if k['name'].startswith('nms')])"; done
