#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r04c4; mkdir -p $OUT; export TMPDIR=/tmp
T="timeout 600"
{ $T python scripts/ab_fps.py rounds2 8x16384x4096 256x16384x4096 512x16384x4096 8x12345x3000 8x16384x1024 8x9000x2000; WS3D_FPS_ROUNDS=1 $T python scripts/ab_fps.py rounds1 8x16384x4096 256x16384x4096 512x16384x4096 8x12345x3000 8x16384x1024 8x9000x2000; } > $OUT/fps_ab.txt 2>&1; cat $OUT/fps_ab.txt
$T python -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_gpu_fullsize.py -m gpu -x -q -k "fps or FPS or sampling" > $OUT/pytest_fps.log 2>&1; tail -15 $OUT/pytest_fps.log
