#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
out=gpurun_out/bq_variants2
mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_independent_search.py tests/test_gpu_fullsize.py -x -q -m gpu -k "ball_query or query_and_group or group or sa1 or c1_gpu or search or binning or config2 or c2" > $out/pytest.log 2>&1; tail -3 $out/pytest.log
python scripts/r06/bq_time.py default 2>&1 | grep -v amdgpu.ids | tee $out/variants.txt
for v in "$@"; do
  WS3D_EXTRA_DEFS="$v" python -m ws3d_amd.build --only ballquery_group.hip > /dev/null 2>$out/build.err || { echo "build failed: $v"; tail -5 $out/build.err; continue; }
  python scripts/r06/bq_time.py "$v" 2>&1 | grep -v amdgpu.ids | tee -a $out/variants.txt
done
python -m ws3d_amd.build --only ballquery_group.hip > /dev/null
