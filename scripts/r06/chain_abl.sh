#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/chain_abl; mkdir -p $OUT
python -m ws3d_amd.build > /dev/null
python scripts/r06/debug_chain.py 2>&1 | tail -6 | tee $OUT/debug.txt
for abl in 0 1 2 4 6 7; do
  WS3D_EXTRA_DEFS="-DWS3D_CHAIN_THREADS=1024 -DWS3D_CHAIN_ABL=$abl" python -m ws3d_amd.build --only chain_mlp.hip > /dev/null
  for kind in hdl64 lidar; do
    echo "== ABL $abl $kind" | tee -a $OUT/abl.txt
    timeout 300 python scripts/r06/bench_chain.py $kind 2>&1 | grep -E "bit-identical|wgs=0|pair\(3\)" | tee -a $OUT/abl.txt
  done
done
python -m ws3d_amd.build --only chain_mlp.hip > /dev/null
