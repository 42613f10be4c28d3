#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/ops_ab; mkdir -p $OUT
python -m ws3d_amd.build > /dev/null
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "group_points_with_the_rows or gather_and_backward or three_interpolate" > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
for b in 8 256; do
  for v in new old; do
    if [ $v = old ]; then export WS3D_GROUP_NO_LDS=1 WS3D_TI_NO_WIDE=1; else unset WS3D_GROUP_NO_LDS WS3D_TI_NO_WIDE; fi
    timeout 600 python bench.py --workload ops --batch $b --steps 40 --warmup 3 --no-cpu-baseline --detail $OUT/ops_${v}_b$b.json 2>/dev/null > /dev/null
    python - $OUT/ops_${v}_b$b.json $v $b <<'PY' | tee -a $OUT/summary.txt
import json, sys
d = json.load(open(sys.argv[1]))
for k in d["kernels"]:
    if k["bound"] == "hbm":
        print("%s batch %s  %-62s %.3f ms  %.2f of 8 TB/s" % (sys.argv[2], sys.argv[3], k["name"][:62], k["ms_per_step"], k["frac_of_8TBps"]))
PY
  done
done
unset WS3D_GROUP_NO_LDS WS3D_TI_NO_WIDE
