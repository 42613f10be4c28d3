#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/fp_ab; mkdir -p $OUT
python -m ws3d_amd.build > /dev/null
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "qinterp or compact_mlp_pair or fast_path_switches or send_rows" > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
timeout 300 python scripts/r06/bench_fp.py hdl64 2>&1 | grep -v amdgpu | tee $OUT/bench_fp.txt
timeout 1200 python scripts/exp_fastpath_ab.py CHAIN_FP False,True 80 2 hdl64 2>&1 | tail -7 | tee $OUT/ab_chain_fp.txt
