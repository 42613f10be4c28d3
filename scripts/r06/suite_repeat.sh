#!/bin/bash
# the GPU suite N times in a row on one box (the once-in-31 bit difference of round 5 stays a watch item): one line per run
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out/repeat
for i in $(seq 1 ${1:-2}); do
  timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/repeat/run_$i.log 2>&1
  echo "run $i: $(tail -1 gpurun_out/repeat/run_$i.log)" | tee -a gpurun_out/repeat/summary.txt
  grep -E "^FAILED|^ERROR" gpurun_out/repeat/run_$i.log | tee -a gpurun_out/repeat/summary.txt
done
