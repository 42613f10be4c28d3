#!/bin/bash
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
O=gpurun_out/r04c21; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "compact or sa1 or sa_mlp" > $O/pytest_compact.log 2>&1; echo "compact rc $?"; tail -2 $O/pytest_compact.log
timeout 900 bash scripts/ubench/sa1_compact_ablation.sh > $O/sa1_compact_ablation.txt 2>&1; echo "rc $?"; grep -v amdgpu.ids $O/sa1_compact_ablation.txt | tail -8
