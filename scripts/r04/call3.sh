#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r04c3; mkdir -p $OUT; export TMPDIR=/tmp
T="timeout 900"
$T python -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_gpu_fullsize.py -m gpu -x -q -k "fps or FPS or sampling" > $OUT/pytest_fps.log 2>&1; tail -3 $OUT/pytest_fps.log
python scripts/ab_fps.py grid64 > $OUT/fps_ab.txt 2>/dev/null; cat $OUT/fps_ab.txt
