#!/bin/bash
# round 4, GPU call 1: the new reference-held parity tests, the 8-rank gloo run, c5 on the new boxes + its traffic pass,
# the c3 traffic pass, and the cold-process bit-stability probe
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r04c1; mkdir -p $OUT; export TMPDIR=/tmp
T="timeout 900"
$T python -m pytest tests/test_golden.py -m gpu -x -q > $OUT/pytest_golden.log 2>&1; tail -3 $OUT/pytest_golden.log
timeout 2400 python -m pytest tests/test_bench_contract.py -m gpu -x -q -k "eight_ranks" > $OUT/pytest_eight.log 2>&1; tail -3 $OUT/pytest_eight.log
cp gpurun_out/eight_rank_c3.json $OUT/ 2>/dev/null
$T python bench.py --workload c5 2>$OUT/bench_c5.err | tail -1 > $OUT/bench_c5_b8.json
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_c5_$c
  $T rocprofv3 --pmc $c --kernel-trace -d /tmp/pmc_c5_$c -o pmc -- python bench.py --workload c5 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmc_c5_$c.log 2>&1
done
python scripts/pmc_traffic.py "$(find /tmp/pmc_c5_FETCH_SIZE -name '*.db' | head -1)" "$(find /tmp/pmc_c5_WRITE_SIZE -name '*.db' | head -1)" \
  $OUT/traffic_c5.json "c5 batch 8 (hdl64, proposals on the visible cars), bytes per launch, rocprofv3 --pmc in separate passes" 8 > /dev/null 2>>$OUT/pmc_c5_WRITE_SIZE.log
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_c3_$c
  $T rocprofv3 --pmc $c --kernel-trace -d /tmp/pmc_c3_$c -o pmc -- python bench.py --workload c3 --pipeline-depth 1 --no-graph --c2-batch 0 --steps 4 --warmup 2 --no-cpu-baseline --no-side-runs > $OUT/pmc_c3_$c.log 2>&1
done
python scripts/pmc_traffic_c3.py "$(find /tmp/pmc_c3_FETCH_SIZE -name '*.db' | head -1)" "$(find /tmp/pmc_c3_WRITE_SIZE -name '*.db' | head -1)" \
  $OUT/traffic_c3.json hdl64 8 > $OUT/traffic_c3_summary.txt 2>>$OUT/pmc_c3_WRITE_SIZE.log
for i in 1 2 3 4 5 6; do $T python scripts/cold_forward_bits.py 4 train; done > $OUT/cold_forward_train.txt 2>&1
for i in 1 2 3; do $T python scripts/cold_forward_bits.py 4 eval; done > $OUT/cold_forward_eval.txt 2>&1
grep RESULT $OUT/cold_forward_*.txt
ls -la $OUT
