#!/bin/bash
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
O=gpurun_out/r04c20; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "compact or sa1 or sa_mlp" > $O/pytest_compact.log 2>&1; echo "compact rc $?"; tail -3 $O/pytest_compact.log
timeout 900 python -m pytest tests/test_golden.py -x -q -m gpu > $O/pytest_golden.log 2>&1; echo "golden rc $?"; tail -2 $O/pytest_golden.log
timeout 900 bash scripts/ubench/sa1_compact_ablation.sh > $O/sa1_compact_ablation.txt 2>&1; echo "rc $?"; grep -v amdgpu.ids $O/sa1_compact_ablation.txt | tail -8
timeout 900 python scripts/exp_fused_compact3.py 80 > $O/exp_fused_compact3.txt 2>&1; grep -v amdgpu.ids $O/exp_fused_compact3.txt | tail -12
