#!/bin/bash
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
O=gpurun_out/r04c34; mkdir -p $O
BQ_DEFS=-DBQC_LARGE_NW=8 timeout 1200 bash scripts/ubench/bq_wide_threshold.sh > $O/bq_wide_threshold.txt 2>&1; grep -v amdgpu.ids $O/bq_wide_threshold.txt | tail -10
