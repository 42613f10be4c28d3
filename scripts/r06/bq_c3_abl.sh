#!/bin/bash
# round 6: which part of the 16 x 4 search launches of a c3 batch costs what (ablation builds; outputs of the ablated builds are wrong by design)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
out=gpurun_out/bq_c3
mkdir -p $out
: > $out/abl.txt
run() { echo "== $1" | tee -a $out/abl.txt; python scripts/ubench/bq_pairs_levels.py 2>&1 | grep -v amdgpu.ids | grep hdl64 | tee -a $out/abl.txt; }
run default
for v in "$@"; do
  WS3D_EXTRA_DEFS="$v" python -m ws3d_amd.build --only ballquery_group.hip > /dev/null 2>$out/build.err || { echo "build failed: $v"; tail -5 $out/build.err; continue; }
  run "$v"
done
python -m ws3d_amd.build --only ballquery_group.hip > /dev/null
