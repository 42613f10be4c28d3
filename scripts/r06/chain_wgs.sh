#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/chain_wgs; mkdir -p $OUT
python -m ws3d_amd.build > /dev/null
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "compact_mlp_pair or compact_pairs_path or fast_path_switches or send_rows" > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
timeout 200 python scripts/r06/bench_chain.py hdl64 2>&1 | tail -10 | tee $OUT/bench_chain.txt
timeout 1200 python scripts/exp_fastpath_ab.py compat.CHAIN_WORKGROUPS 0,128,64,32 80 2 hdl64 2>&1 | tail -14 | tee $OUT/ab_wgs.txt
