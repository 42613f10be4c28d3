#!/bin/bash
# round 6: the tile-walking forms of ws3d_qinterp_gemm and ws3d_compact_mlp_pair kinds 2 / 1 -- parity, then the workgroup caps on the 20-deep step
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/tune2; mkdir -p $OUT
python -m ws3d_amd.build > /dev/null
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "qinterp or compact_mlp_pair or fast_path_switches or zero_arena or compact_pairs_path" > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
timeout 1200 python scripts/exp_fastpath_ab.py tune:fp_wgs 0,1024,768,512 80 2 hdl64 2>&1 | tail -13 | tee $OUT/ab_fp_wgs.txt
timeout 1200 python scripts/exp_fastpath_ab.py tune:pair_wgs 0,1024,512,256 80 2 hdl64 2>&1 | tail -13 | tee $OUT/ab_pair_wgs.txt
timeout 900 python scripts/exp_fastpath_ab.py tune:mlp2_wgs 0,128 80 3 hdl64 2>&1 | tail -9 | tee $OUT/ab_mlp2_wgs.txt
