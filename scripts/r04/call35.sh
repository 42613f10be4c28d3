#!/bin/bash
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
O=gpurun_out/r04c35; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "qinterp" > $O/pytest_qinterp.log 2>&1; echo "qinterp rc $?"; tail -4 $O/pytest_qinterp.log
for rows in 100000 30000 1; do
timeout 900 python scripts/exp_fastpath_ab.py FUSED_QINTERP_GEMM_MIN_ROWS 1152921504606846976 $rows 80 > $O/exp_qinterp_gemm_$rows.txt 2>&1; grep -v amdgpu.ids $O/exp_qinterp_gemm_$rows.txt | grep hdl64
done
