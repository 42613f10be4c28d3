#!/bin/bash
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
O=gpurun_out/r04c17; mkdir -p $O
timeout 600 python scripts/exp_bin_ahead.py > $O/exp_bin_ahead.txt 2>&1; echo "rc $?"; grep -v amdgpu.ids $O/exp_bin_ahead.txt | tail -14
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "sampling_plan or consecutive_forward or recycled or fastpath or stage1" > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
