#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r04c5; mkdir -p $OUT; export TMPDIR=/tmp
T="timeout 600"
{ $T python scripts/ab_fps.py rounds2 8x16384x4096 512x16384x4096 8x12345x3000 8x16384x1024; } > $OUT/fps_ab.txt 2>&1; cat $OUT/fps_ab.txt
bash scripts/ubench/fps_rounds2_prof.sh 2>&1 | tee $OUT/fps_rounds2_segments.txt
$T python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fps_bit_exact or fps_ties" > $OUT/pytest_fps.log 2>&1; tail -3 $OUT/pytest_fps.log
