#!/bin/bash
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
O=gpurun_out/r04c22; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "qinterp" > $O/pytest_qinterp.log 2>&1; echo "qinterp rc $?"; tail -5 $O/pytest_qinterp.log
timeout 900 python -m pytest tests/test_golden.py -x -q -m gpu > $O/pytest_golden.log 2>&1; echo "golden rc $?"; tail -2 $O/pytest_golden.log
timeout 900 python scripts/exp_fastpath_ab.py FUSED_QINTERP_GEMM False True 80 > $O/exp_qinterp_gemm.txt 2>&1; grep -v amdgpu.ids $O/exp_qinterp_gemm.txt | tail -10
