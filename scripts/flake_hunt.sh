#!/bin/bash
# scripts/flake_hunt.sh <outdir> [train iterations] [eval iterations]: the hunt in every mode, one log
OUT=${1:-gpurun_out/flake}; NT=${2:-500}; NE=${3:-500}
mkdir -p $OUT
L=$OUT/flake_hunt.txt
: > $L
for m in "plain" "churn" "churn busy"; do timeout 1500 python scripts/flake_hunt.py train $NT $m >> $L 2>&1; done
for m in "side" "one" "sync" "side busy" "sync busy"; do timeout 1500 python scripts/flake_hunt.py eval $NE $m >> $L 2>&1; done
grep -c MISMATCH $L; grep RESULT $L
