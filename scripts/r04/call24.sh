#!/bin/bash
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
O=gpurun_out/r04c24; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_golden.py -x -q -m gpu -k "ball or query or group or fullsize or compact or golden or stage1 or pairs" > $O/pytest_bq.log 2>&1; echo "bq rc $?"; tail -3 $O/pytest_bq.log
echo "## old library (4 waves x 16 centres everywhere)" > $O/bq_pairs_levels.txt
WS3D_HIP_LIB=$PWD/ws3d_amd/libws3d_hip_old.so timeout 600 python scripts/ubench/bq_pairs_levels.py >> $O/bq_pairs_levels.txt 2>&1
echo "## new library (16 waves x 4 centres below 2048 tiles)" >> $O/bq_pairs_levels.txt
timeout 600 python scripts/ubench/bq_pairs_levels.py >> $O/bq_pairs_levels.txt 2>&1
grep -v amdgpu.ids $O/bq_pairs_levels.txt
timeout 600 python scripts/exp_latency_segments.py 30 > $O/latency_segments.txt 2>&1; grep BIN_INPUT $O/latency_segments.txt
