#!/bin/bash
# kernel statistics of the reference-layout leg -> gpurun_out/reflayout/
set -u
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
out=gpurun_out/reflayout
mkdir -p $out
python scripts/r06/reflayout_prof.py 20 > $out/plain.txt 2>&1
tail -1 $out/plain.txt
rm -rf /tmp/rl_prof
rocprofv3 --kernel-trace --stats -d /tmp/rl_prof -o rl -- python scripts/r06/reflayout_prof.py 20 > $out/prof.log 2>&1
python scripts/rocpd_stats.py "$(find /tmp/rl_prof -name '*.db' | head -1)" > $out/reflayout_kernel_stats.csv 2>>$out/prof.log
head -45 $out/reflayout_kernel_stats.csv | cut -c1-200
