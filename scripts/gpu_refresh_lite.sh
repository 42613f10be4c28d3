#!/bin/bash
# The evidence that has to match the code state (~25 GPU-minutes): GPU test suite, kernel stats of c2 / c3 / c5, the default lines, the c2
# batch sweep, the ops workload, launches per step, marginal costs -- and LAST the counter passes bench.py quotes
# (profiles/traffic*.json carry the blob hashes of the sources they describe: taken last, they describe the code that is committed).
# Everything lands in gpurun_out/refresh/; scripts/copy_refresh.sh <round> copies it under profiles/.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/refresh
mkdir -p $OUT
export TMPDIR=/tmp
T="timeout 900"
python -m ws3d_amd.build --all-dist-modes > /dev/null
$T python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; tail -3 $OUT/pytest_gpu.log
for w in c2 c3 c5; do
  extra=""; [ $w = c3 ] && extra="--pipeline-depth 1 --no-graph --c2-batch 0"
  rm -rf /tmp/prof_$w
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$w -o $w -- python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline $extra > $OUT/prof_$w.log 2>&1
  python scripts/rocpd_stats.py "$(find /tmp/prof_$w -name '*.db' | head -1)" > $OUT/${w}_kernel_stats.csv 2>>$OUT/prof_$w.log
done
$T python bench.py --steps 20 --warmup 5 --detail $OUT/bench_default_steps20_warmup5_detail.json 2>$OUT/bench_default.err > $OUT/bench_default_steps20_warmup5.json
$T python bench.py --detail $OUT/bench_default_detail.json 2>>$OUT/bench_default.err > $OUT/bench_default_line.json
$T python bench.py --full-line --no-cpu-baseline --no-side-runs --c2-batch 0 --steps 160 2>>$OUT/bench_default.err | tail -1 > $OUT/bench_c3_b8_steps160.json
for b in 1 8 64 256 512 1024; do          # the batch sweep of the c2 path, with every number
  $T python bench.py --full-line --workload c2 --batch $b --no-cpu-baseline 2>>$OUT/bench_default.err | tail -1 > $OUT/bench_c2_b$b.json
done
$T python bench.py --full-line --workload c5 2>>$OUT/bench_default.err | tail -1 > $OUT/bench_c5_b8.json
for b in 8 256; do
  $T python bench.py --workload ops --batch $b --detail $OUT/bench_ops_b${b}_detail.json 2>>$OUT/bench_default.err > $OUT/bench_ops_b$b.json
done
timeout 900 bash scripts/launches_per_step.sh > $OUT/launches_per_step.txt 2>&1
timeout 1500 bash scripts/throughput_marginal.sh > $OUT/throughput_marginal_cost.txt 2>&1
timeout 600 bash scripts/r06/cu_time_budget.sh > $OUT/cu_time_budget.log 2>&1; cp gpurun_out/cu/cu_time_budget.txt $OUT/cu_time_budget.txt 2>/dev/null
timeout 300 python scripts/r06/bq_time.py default 2>&1 | grep -v amdgpu.ids > $OUT/bq_time.txt
# ---- counters, LAST
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c /tmp/pmc_c5_$c /tmp/pmc_c3_$c /tmp/pmc_ops8_$c /tmp/pmc_ops256_$c
  timeout 300 rocprofv3 --pmc $c --kernel-trace -d /tmp/pmc_$c -o pmc -- python bench.py --workload c2 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmc_$c.log 2>&1
  timeout 300 rocprofv3 --pmc $c --kernel-trace -d /tmp/pmc_c5_$c -o pmc -- python bench.py --workload c5 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmc_c5_$c.log 2>&1
  timeout 300 rocprofv3 --pmc $c --kernel-trace -d /tmp/pmc_c3_$c -o pmc -- python bench.py --workload c3 --pipeline-depth 1 --no-graph --c2-batch 0 --steps 4 --warmup 2 --no-cpu-baseline --no-side-runs > $OUT/pmc_c3_$c.log 2>&1
  timeout 300 rocprofv3 --pmc $c --kernel-trace -d /tmp/pmc_ops8_$c -o pmc -- python bench.py --workload ops --batch 8 --steps 4 --warmup 1 --no-cpu-baseline > $OUT/pmc_ops8_$c.log 2>&1
  timeout 300 rocprofv3 --pmc $c --kernel-trace -d /tmp/pmc_ops256_$c -o pmc -- python bench.py --workload ops --batch 256 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmc_ops256_$c.log 2>&1
done
db() { find /tmp/$1 -name '*.db' | head -1; }
python scripts/pmc_traffic.py "$(db pmc_FETCH_SIZE)" "$(db pmc_WRITE_SIZE)" $OUT/traffic.json "c2 batch 512, bytes per launch, rocprofv3 --pmc in separate passes" 512 > /dev/null 2>>$OUT/pmc_WRITE_SIZE.log
python scripts/pmc_traffic.py "$(db pmc_c5_FETCH_SIZE)" "$(db pmc_c5_WRITE_SIZE)" $OUT/traffic_c5.json "c5 batch 8, bytes per launch, rocprofv3 --pmc in separate passes" 8 > /dev/null 2>>$OUT/pmc_c5_WRITE_SIZE.log
python scripts/pmc_traffic_c3.py "$(db pmc_c3_FETCH_SIZE)" "$(db pmc_c3_WRITE_SIZE)" $OUT/traffic_c3.json hdl64 8 > $OUT/traffic_c3_summary.txt 2>>$OUT/pmc_c3_WRITE_SIZE.log
python scripts/pmc_traffic_ops.py "$(db pmc_ops8_FETCH_SIZE)" "$(db pmc_ops8_WRITE_SIZE)" $OUT/traffic_ops.json 8 > /dev/null 2>>$OUT/pmc_ops8_WRITE_SIZE.log
python scripts/pmc_traffic_ops.py "$(db pmc_ops256_FETCH_SIZE)" "$(db pmc_ops256_WRITE_SIZE)" $OUT/traffic_ops256.json 256 > /dev/null 2>>$OUT/pmc_ops256_WRITE_SIZE.log
$T bash scripts/pmc_fps_valu.sh $OUT/pmc_fps > $OUT/pmc_fps_valu.txt 2>&1
timeout 600 bash scripts/r06/bq_emit_instr.sh $OUT > $OUT/bq_emit_instr_per_byte.txt 2>&1
cp $OUT/traffic.json $OUT/traffic_c5.json $OUT/traffic_c3.json $OUT/traffic_ops.json $OUT/traffic_ops256.json profiles/ 2>/dev/null
[ -s $OUT/pmc_fps/traffic_fps_valu.json ] && cp $OUT/pmc_fps/traffic_fps_valu.json profiles/
# the lines once more, now quoting the fresh counter files
$T python bench.py --steps 20 --warmup 5 --detail $OUT/bench_default_steps20_warmup5_detail.json 2>$OUT/bench_default2.err > $OUT/bench_default_steps20_warmup5.json
$T python bench.py --full-line --workload c5 2>>$OUT/bench_default2.err | tail -1 > $OUT/bench_c5_b8.json
$T python bench.py --full-line --workload c2 --batch 512 2>>$OUT/bench_default2.err | tail -1 > $OUT/bench_c2_b512.json
for b in 8 256; do
  $T python bench.py --workload ops --batch $b --no-cpu-baseline --detail $OUT/bench_ops_b${b}_detail.json 2>>$OUT/bench_default2.err > $OUT/bench_ops_b$b.json
done
cat $OUT/bench_default_steps20_warmup5.json
