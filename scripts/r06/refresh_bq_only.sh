#!/bin/bash
# round 6, last change (ws3d_tune key 5: only ballquery_group.hip / pipeline.py / compat.py changed): the tests behind the point where the
# suite stopped, then the counter passes whose source lists name ballquery_group.hip (c2, c3, ops) and the lines that quote them
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/refresh; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_independent_search.py tests/test_kitti_io.py tests/test_train.py tests/test_oracle_vs_ref.py tests/test_oracle_properties.py -q -m gpu \
  -k "eight_waves or fill_equals or send_rows or staged_in_lds or zero_arena or test_independent_search or test_kitti_io or test_train or test_oracle" > $OUT/pytest_tail.log 2>&1; tail -2 $OUT/pytest_tail.log
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c /tmp/pmc_c3_$c /tmp/pmc_ops8_$c /tmp/pmc_ops256_$c
  timeout 300 rocprofv3 --pmc $c --kernel-trace -d /tmp/pmc_$c -o pmc -- python bench.py --workload c2 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmc_$c.log 2>&1
  timeout 300 rocprofv3 --pmc $c --kernel-trace -d /tmp/pmc_c3_$c -o pmc -- python bench.py --workload c3 --pipeline-depth 1 --no-graph --c2-batch 0 --steps 4 --warmup 2 --no-cpu-baseline --no-side-runs > $OUT/pmc_c3_$c.log 2>&1
  timeout 300 rocprofv3 --pmc $c --kernel-trace -d /tmp/pmc_ops8_$c -o pmc -- python bench.py --workload ops --batch 8 --steps 4 --warmup 1 --no-cpu-baseline > $OUT/pmc_ops8_$c.log 2>&1
  timeout 300 rocprofv3 --pmc $c --kernel-trace -d /tmp/pmc_ops256_$c -o pmc -- python bench.py --workload ops --batch 256 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmc_ops256_$c.log 2>&1
done
db() { find /tmp/$1 -name '*.db' | head -1; }
python scripts/pmc_traffic.py "$(db pmc_FETCH_SIZE)" "$(db pmc_WRITE_SIZE)" $OUT/traffic.json "c2 batch 512, bytes per launch, rocprofv3 --pmc in separate passes" 512 > /dev/null 2>>$OUT/pmc_WRITE_SIZE.log
python scripts/pmc_traffic_c3.py "$(db pmc_c3_FETCH_SIZE)" "$(db pmc_c3_WRITE_SIZE)" $OUT/traffic_c3.json hdl64 8 > $OUT/traffic_c3_summary.txt 2>>$OUT/pmc_c3_WRITE_SIZE.log
python scripts/pmc_traffic_ops.py "$(db pmc_ops8_FETCH_SIZE)" "$(db pmc_ops8_WRITE_SIZE)" $OUT/traffic_ops.json 8 > /dev/null 2>>$OUT/pmc_ops8_WRITE_SIZE.log
python scripts/pmc_traffic_ops.py "$(db pmc_ops256_FETCH_SIZE)" "$(db pmc_ops256_WRITE_SIZE)" $OUT/traffic_ops256.json 256 > /dev/null 2>>$OUT/pmc_ops256_WRITE_SIZE.log
cp $OUT/traffic.json $OUT/traffic_c3.json $OUT/traffic_ops.json $OUT/traffic_ops256.json profiles/
T="timeout 600"
$T python bench.py --steps 20 --warmup 5 --detail $OUT/bench_default_steps20_warmup5_detail.json 2>$OUT/bench_default2.err > $OUT/bench_default_steps20_warmup5.json
$T python bench.py --detail $OUT/bench_default_detail.json 2>>$OUT/bench_default2.err > $OUT/bench_default_line.json
$T python bench.py --full-line --workload c2 --batch 512 2>>$OUT/bench_default2.err | tail -1 > $OUT/bench_c2_b512.json
for b in 8 256; do
  $T python bench.py --workload ops --batch $b --no-cpu-baseline --detail $OUT/bench_ops_b${b}_detail.json 2>>$OUT/bench_default2.err > $OUT/bench_ops_b$b.json
done
timeout 300 bash scripts/launches_per_step.sh > $OUT/launches_per_step.txt 2>&1
python -c "
import json
for f in ('bench_default_steps20_warmup5.json','bench_default_line.json'):
    d=json.load(open('$OUT/'+f)); print(f, d['value'], d['value_steady'], d['roofline']['traffic'], d['c2_query_group_hbm_frac'])"
