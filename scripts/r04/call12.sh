#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r04c12; mkdir -p $OUT; export TMPDIR=/tmp
for i in 1 2 3; do
  timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest_gpu_$i.log 2>&1; tail -4 $OUT/pytest_gpu_$i.log | tr '\n' ' '; echo
done
grep -l "FAILED\|failed" $OUT/*.log
