#!/bin/bash
# hunting the once-in-~17-runs difference of test_sampling_plan_equals_sampling_inside_the_modules: hot GPU (a fuzzer first, as in the run that failed), then the parity file over and over
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
O=gpurun_out/r04c31; mkdir -p $O
timeout 200 python scripts/fuzz_compact.py --seconds 90 --seed 7 > $O/fuzz.txt 2>&1
for i in $(seq 1 14); do
  timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x > $O/parity_$i.log 2>&1; rc=$?
  echo "run $i rc $rc: $(tail -1 $O/parity_$i.log)"
  if [ $rc -ne 0 ]; then grep -h "differs from the clone\|both passes still equal\|first module whose\|AssertionError\|FAILED" $O/parity_$i.log | cut -c1-900; fi
done
