#!/bin/bash
# Issued VALU instructions of the FPS kernels from the hardware counters (separate rocprofv3 --pmc passes, kernel trace only):
#   SQ_INSTS_VALU per launch -> wave64 VALU instructions; / duration / (1024 SIMDs x 2.4e9 / 2 clk) = measured share of the VALU
#   issue roof, next to bench.py's model figure (8 per point and step).  GRBM_GUI_ACTIVE / duration = effective clock.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=${1:-gpurun_out/pmc_valu}; mkdir -p $OUT; export TMPDIR=/tmp
for b in 512 256; do
  for c in SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES; do
    rm -rf /tmp/pv_$c
    (cd /tmp && timeout 600 rocprofv3 --pmc $c --kernel-trace -d /tmp/pv_$c -o pv -- python $OLDPWD/bench.py --workload c2 --batch $b --steps 3 --warmup 1 --no-cpu-baseline > $OLDPWD/$OUT/log_${b}_$c.txt 2>&1)
    python - "$(find /tmp/pv_$c -name '*.db' | head -1)" $c $b >> $OUT/valu_counters.txt <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); c, b = sys.argv[2], sys.argv[3]
try:
    rows = db.execute("select kernel_name, avg(value), count(*) from counters_collection where counter_name=? group by kernel_name", (c,)).fetchall()
except Exception as e:
    rows = []; print("batch", b, c, "ERROR", e)
dur = {r[0]: r[1] for r in db.execute("select name, avg(end-start) from kernels group by name").fetchall()}
for name, v, n in rows:
    if "fps" in name:
        print("batch %s  %-16s %-70s avg %.6g per launch (%d launches), traced duration %.3f ms" % (b, c, name[:70], v, n, dur.get(name, 0) / 1e6))
PY
  done
done
cat $OUT/valu_counters.txt
