#!/bin/bash
# round 6: the register-chained SA2 kernel -- parity test, isolated timing of build variants, throughput-mode ABAB
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/chain_ab; mkdir -p $OUT
python -m ws3d_amd.build > /dev/null
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "compact_mlp_pair or compact_pairs_path or fast_path_switches" > $OUT/pytest.log 2>&1; tail -5 $OUT/pytest.log
for v in "768 1" "1024 1" "512 2"; do
  set -- $v
  WS3D_EXTRA_DEFS="-DWS3D_CHAIN_THREADS=$1 -DWS3D_CHAIN_G3=$2" python -m ws3d_amd.build --only chain_mlp.hip > /dev/null
  echo "== threads $1 G3 $2" | tee -a $OUT/bench_chain.txt
  timeout 300 python scripts/r06/bench_chain.py hdl64 2>&1 | tail -12 | tee -a $OUT/bench_chain.txt
done
python -m ws3d_amd.build --only chain_mlp.hip > /dev/null       # back to the default build
timeout 300 python scripts/r06/bench_chain.py lidar 2>&1 | tail -12 | tee -a $OUT/bench_chain.txt
timeout 900 python scripts/exp_fastpath_ab.py CHAIN_MLP False True 80 2 hdl64 2>&1 | tail -8 | tee $OUT/ab_chain.txt
