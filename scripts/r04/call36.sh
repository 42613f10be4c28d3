#!/bin/bash
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
O=gpurun_out/r04c36; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_bench_contract.py -x -q -m gpu -k "switches or qinterp or golden or stage1 or fast or pipeline or default_line or recycled or consecutive or kitti" > $O/pytest.log 2>&1; echo "rc $?"; tail -3 $O/pytest.log
