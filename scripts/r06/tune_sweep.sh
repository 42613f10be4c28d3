#!/bin/bash
# round 6: launch-geometry sweeps of the persistent kernels on the 20-deep c3 step (ABAB per knob, hdl64)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/tune; mkdir -p $OUT
python -m ws3d_amd.build > /dev/null
timeout 900 python scripts/exp_fastpath_ab.py tune:mlp2_wgs 0,128,64 80 2 hdl64 2>&1 | tail -10 | tee $OUT/ab_mlp2_wgs.txt
timeout 900 python scripts/exp_fastpath_ab.py tune:sa1_wgs 0,384,192,96 80 2 hdl64 2>&1 | tail -13 | tee $OUT/ab_sa1_wgs.txt
timeout 900 python scripts/exp_fastpath_ab.py tune:chain_wgs 64,96,48,256 80 2 hdl64 2>&1 | tail -13 | tee $OUT/ab_chain_wgs.txt
