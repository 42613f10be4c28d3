#!/bin/bash
# round 6: the searches of one c3 batch (ball_query_pairs per level and scale, timed alone) and the 20-deep c3 line, A/B builds of ballquery_group.hip
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
out=gpurun_out/bq_c3
mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_independent_search.py -x -q -m gpu -k "ball_query or query_and_group or group or sa1 or c1_gpu or search or headline or stage1" > $out/pytest.log 2>&1; tail -3 $out/pytest.log
run() {
  echo "== $1" | tee -a $out/ab.txt
  python scripts/ubench/bq_pairs_levels.py 2>&1 | grep -v amdgpu.ids | tee -a $out/ab.txt
  python bench.py --full-line --no-cpu-baseline --no-side-runs --c2-batch 0 --steps 160 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c3 160 steps: %.1f scenes/s  %.4f ms/step' % (d['value'], d['ms_per_step']))" | tee -a $out/ab.txt
}
: > $out/ab.txt
run default
for v in "$@"; do
  WS3D_EXTRA_DEFS="$v" python -m ws3d_amd.build --only ballquery_group.hip > /dev/null 2>$out/build.err || { echo "build failed: $v"; tail -5 $out/build.err; continue; }
  run "$v"
done
python -m ws3d_amd.build --only ballquery_group.hip > /dev/null
run "default again"
