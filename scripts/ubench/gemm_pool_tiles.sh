#!/bin/bash
# gemm_pool with (64 MB) x (64 NB) output tiles (WS3D_GP_TILE = MB NB) on the six last-layer shapes of SA2..SA4, and the
# matrix-core / LDS / L2 counters of the 64 x 64 and the 128 x 128 kernels (separate --pmc passes, kernel trace only).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=${1:-gpurun_out/gp_tiles}; mkdir -p $OUT; export TMPDIR=/tmp; export PYTHONPATH=$PWD
for t in 11 21 12 22 41 42; do
  echo "== WS3D_GP_TILE=$t" >> $OUT/timing.txt
  WS3D_GP_TILE=$t timeout 300 python scripts/ubench/gemm_pool.py 2>&1 | grep -v amdgpu.ids >> $OUT/timing.txt
done
cat $OUT/timing.txt
for t in 11 22; do
  for c in SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU; do
    rm -rf /tmp/gp_$c
    (cd /tmp && WS3D_GP_TILE=$t timeout 300 rocprofv3 --pmc $c --kernel-trace -d /tmp/gp_$c -o gp -- python $OLDPWD/scripts/ubench/gemm_pool.py > $OLDPWD/$OUT/log_$c.txt 2>&1)
    python - "$(find /tmp/gp_$c -name '*.db' | head -1)" $c $t >> $OUT/counters.txt <<'PY'
import sqlite3, sys
c, t = sys.argv[2], sys.argv[3]
try:
    db = sqlite3.connect(sys.argv[1])
    rows = db.execute("select kernel_name, avg(value), count(*) from counters_collection where counter_name=? group by kernel_name", (c,)).fetchall()
    dur = {r[0]: r[1] for r in db.execute("select name, avg(end-start) from kernels group by name").fetchall()}
    for name, v, n in rows:
        if "gemm_pool" in name:
            print("tile %s  %-28s %-40s avg %.6g per launch (%d), %.1f us" % (t, c, name[6:46], v, n, dur.get(name, 0) / 1e3))
except Exception as e:
    print("tile", t, c, "ERROR", e)
PY
  done
done
cat $OUT/counters.txt
