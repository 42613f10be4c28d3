#!/bin/bash
# The evidence that has to match the code state (a subset of scripts/gpu_refresh.sh, ~20 GPU-minutes): GPU test suite, kernel stats of
# c2 / c3 / c5, the HBM counter passes bench.py quotes (profiles/traffic*.json), the sampling kernel's VALU count, the default lines.
# Everything lands in gpurun_out/refresh/; scripts/copy_refresh.sh <round> copies it under profiles/.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/refresh
mkdir -p $OUT
export TMPDIR=/tmp
T="timeout 900"
$T python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; tail -3 $OUT/pytest_gpu.log
for w in c2 c3 c5; do
  extra=""; [ $w = c3 ] && extra="--pipeline-depth 1 --no-graph --c2-batch 0"
  rm -rf /tmp/prof_$w
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$w -o $w -- python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline $extra > $OUT/prof_$w.log 2>&1
  python scripts/rocpd_stats.py "$(find /tmp/prof_$w -name '*.db' | head -1)" > $OUT/${w}_kernel_stats.csv 2>>$OUT/prof_$w.log
done
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c /tmp/pmc_c5_$c /tmp/pmc_c3_$c
  timeout 300 rocprofv3 --pmc $c --kernel-trace -d /tmp/pmc_$c -o pmc -- python bench.py --workload c2 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmc_$c.log 2>&1
  timeout 300 rocprofv3 --pmc $c --kernel-trace -d /tmp/pmc_c5_$c -o pmc -- python bench.py --workload c5 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmc_c5_$c.log 2>&1
  timeout 300 rocprofv3 --pmc $c --kernel-trace -d /tmp/pmc_c3_$c -o pmc -- python bench.py --workload c3 --pipeline-depth 1 --no-graph --c2-batch 0 --steps 4 --warmup 2 --no-cpu-baseline --no-side-runs > $OUT/pmc_c3_$c.log 2>&1
done
python scripts/pmc_traffic.py "$(find /tmp/pmc_FETCH_SIZE -name '*.db' | head -1)" "$(find /tmp/pmc_WRITE_SIZE -name '*.db' | head -1)" \
  $OUT/traffic.json "c2 batch 512, bytes per launch, rocprofv3 --pmc in separate passes" 512 > /dev/null 2>>$OUT/pmc_WRITE_SIZE.log
python scripts/pmc_traffic.py "$(find /tmp/pmc_c5_FETCH_SIZE -name '*.db' | head -1)" "$(find /tmp/pmc_c5_WRITE_SIZE -name '*.db' | head -1)" \
  $OUT/traffic_c5.json "c5 batch 8, bytes per launch, rocprofv3 --pmc in separate passes" 8 > /dev/null 2>>$OUT/pmc_c5_WRITE_SIZE.log
python scripts/pmc_traffic_c3.py "$(find /tmp/pmc_c3_FETCH_SIZE -name '*.db' | head -1)" "$(find /tmp/pmc_c3_WRITE_SIZE -name '*.db' | head -1)" \
  $OUT/traffic_c3.json hdl64 8 > $OUT/traffic_c3_summary.txt 2>>$OUT/pmc_c3_WRITE_SIZE.log
$T bash scripts/pmc_fps_valu.sh $OUT/pmc_fps > $OUT/pmc_fps_valu.txt 2>&1
cp $OUT/traffic.json $OUT/traffic_c5.json $OUT/traffic_c3.json profiles/ 2>/dev/null
[ -s $OUT/pmc_fps/traffic_fps_valu.json ] && cp $OUT/pmc_fps/traffic_fps_valu.json profiles/
$T python bench.py --steps 20 --warmup 5 --detail $OUT/bench_default_steps20_warmup5_detail.json 2>$OUT/bench_default.err > $OUT/bench_default_steps20_warmup5.json
$T python bench.py --detail $OUT/bench_default_detail.json 2>>$OUT/bench_default.err > $OUT/bench_default_line.json
$T python bench.py --full-line --no-cpu-baseline --no-side-runs --c2-batch 0 --steps 160 2>>$OUT/bench_default.err | tail -1 > $OUT/bench_c3_b8_steps160.json
$T python bench.py --full-line --workload c2 2>>$OUT/bench_default.err | tail -1 > $OUT/bench_c2_b512.json
$T python bench.py --full-line --workload c5 2>>$OUT/bench_default.err | tail -1 > $OUT/bench_c5_b8.json
cat $OUT/bench_default_steps20_warmup5.json
