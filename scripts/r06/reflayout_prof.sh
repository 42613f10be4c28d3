#!/bin/bash
# kernel statistics of the reference-layout leg -> gpurun_out/reflayout/
set -u
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
out=gpurun_out/reflayout
mkdir -p $out
python scripts/r06/reflayout_prof.py 20 > $out/plain.txt 2>&1
tail -1 $out/plain.txt
rocprofv3 --kernel-trace --stats -d $out/prof -o rl -- python scripts/r06/reflayout_prof.py 20 > $out/prof.log 2>&1
f=$(find $out/prof -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp "$f" $out/reflayout_kernel_stats.csv && head -40 $out/reflayout_kernel_stats.csv | cut -c1-220
