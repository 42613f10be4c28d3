#!/bin/bash
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
O=gpurun_out/r04c15; mkdir -p $O
timeout 1500 python scripts/exp_tunable_gemm.py 80 $O/tunableop_results.csv > $O/exp_tunable_gemm.txt 2>&1; echo "rc $?"
cat $O/exp_tunable_gemm.txt | tail -60
