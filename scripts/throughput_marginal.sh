#!/bin/bash
# What each launch family costs in THROUGHPUT mode (20 batches in flight): bench.py with that family's launches issued twice
# (WS3D_BENCH_DOUBLE, bench_c3.py) -- the rise of ms_per_step over the plain run is the family's marginal cost when everything
# else of 19 other batches runs beside it.  Families whose rise is ~0 hide behind other work; the rest bound the headline.
cd "$(dirname "$0")/.."
KIND=${1:-hdl64}
run() { WS3D_BENCH_DOUBLE="$1" python bench.py --full-line --kind $KIND --steps 60 --warmup 3 --no-side-runs --no-cpu-baseline --c2-batch 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1])
print('%-58s %8.4f ms/batch  latency %6.3f ms' % (sys.argv[1] or '(plain)', d['ms_per_step'], d['latency_mode']['ms_per_batch']))" "$1"; }
run ""
for fam in "prologue" "fps level 1" "fps levels 2-4" "binning" "ball_query" "SharedMLP SA1" "SharedMLP SA2-4 both scales" "three_nn" "FP first layer" "heads" "proposals" "nms" "roipool3d" "library"; do run "$fam"; done
run ""
