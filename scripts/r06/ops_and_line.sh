#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/ops; mkdir -p $OUT
export TMPDIR=/tmp
python -m ws3d_amd.build > /dev/null
timeout 900 python -m pytest tests/test_golden.py -x -q -m gpu -k "headline" > $OUT/pytest_headline.log 2>&1; tail -4 $OUT/pytest_headline.log
for b in 8 256; do
  timeout 600 python bench.py --workload ops --batch $b --steps 40 --warmup 3 --detail $OUT/bench_ops_b${b}_detail.json 2>$OUT/bench_ops_b$b.err > $OUT/bench_ops_b$b.json; tail -c 1500 $OUT/bench_ops_b$b.json
done
timeout 900 python bench.py --steps 20 --warmup 5 --detail $OUT/bench_default_detail.json 2>$OUT/bench_default.err > $OUT/bench_default.json; cat $OUT/bench_default.json
