#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r04c8; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; tail -8 $OUT/pytest_gpu.log
timeout 900 python bench.py --no-cpu-baseline --no-side-runs 2>/dev/null | tail -1 > $OUT/bench.json
python -c "
import json; d=json.load(open('$OUT/bench.json')); print('value %.0f' % d['value'], 'latency %.3f' % d['latency_mode']['ms_per_batch'], 'c2 %.0f' % d['c2']['scenes_per_s_per_gpu'])"
