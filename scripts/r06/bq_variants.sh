#!/bin/bash
# round 6: what bounds the fused query + group kernel of the c2 block?  the store pattern alone, then A/B builds of the kernel
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
out=gpurun_out/bq_variants
mkdir -p $out
hipcc -O3 --offload-arch=gfx950 -Wno-unused-value scripts/ubench/group_store_pattern.hip -o /tmp/gsp && /tmp/gsp > $out/store_pattern.txt 2>&1
cat $out/store_pattern.txt
python scripts/r06/bq_time.py default 2>&1 | grep -v amdgpu.ids | tee $out/variants.txt
for v in "-DWS3D_BQ_NO_NT" "-DBQC_LARGE_NW=8" "-DBQC_LARGE_NW=16" "-DBQC_LARGE_NW=2" "-DWS3D_BQC_NO_SEARCH" "-DWS3D_BQC_NO_EMIT" "-DWS3D_BQC_NO_SEARCH -DWS3D_BQ_NO_NT"; do
  WS3D_EXTRA_DEFS="$v" python -m ws3d_amd.build --only ballquery_group.hip > /dev/null 2>$out/build.err || { echo "build failed: $v"; tail -5 $out/build.err; continue; }
  python scripts/r06/bq_time.py "$v" 2>&1 | grep -v amdgpu.ids | tee -a $out/variants.txt
done
python -m ws3d_amd.build --only ballquery_group.hip > /dev/null
