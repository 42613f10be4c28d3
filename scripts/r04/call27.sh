#!/bin/bash
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
O=gpurun_out/r04c27; mkdir -p $O
timeout 600 python scripts/fuzz_parity.py --seconds 240 --seed 41 > $O/fuzz_parity.txt 2>&1; echo "fuzz_parity rc $?"; tail -3 $O/fuzz_parity.txt
timeout 400 python scripts/fuzz_compact.py --seconds 180 --seed 42 > $O/fuzz_compact.txt 2>&1; echo "fuzz_compact rc $?"; tail -3 $O/fuzz_compact.txt
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -2 $O/pytest_gpu.log
