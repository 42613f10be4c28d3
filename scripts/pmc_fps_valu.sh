#!/bin/bash
# Issued VALU instructions of the level-1 sampling kernel from the hardware counters (separate rocprofv3 --pmc passes, kernel trace
# only): SQ_INSTS_VALU per launch = wave64 VALU instructions, x 64 lanes / duration / (1024 SIMDs x 32 lanes/clk x 2.4 GHz) = the
# PHYSICAL share of the chip's VALU issue roof; SQ_BUSY_CYCLES / GRBM_GUI_ACTIVE for the clock.  bench.py's c2 workload at the
# batch sizes of the default line (512 scenes: the c2 block; 8 scenes: level 1 of the c3 step), default generator.
#   -> <out>/traffic_fps_valu.json  (copy to profiles/: bench.py reads profiles/traffic_fps_valu.json)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=${1:-gpurun_out/pmc_fps}; mkdir -p $OUT; export TMPDIR=/tmp
KINDS=${2:-"hdl64 lidar"}
rm -f $OUT/fps_valu_rows.txt
for KIND in $KINDS; do
for b in 512 8; do
  for c in SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE; do
    rm -rf /tmp/pv_$c
    (cd /tmp && timeout 200 rocprofv3 --pmc $c --kernel-trace -d /tmp/pv_$c -o pv -- python $OLDPWD/bench.py --workload c2 --kind $KIND --batch $b --steps 3 --warmup 1 --no-cpu-baseline > $OLDPWD/$OUT/log_${b}_$c.txt 2>&1)
    python - "$(find /tmp/pv_$c -name '*.db' | head -1)" $c $b $KIND >> $OUT/fps_valu_rows.txt <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); c, b = sys.argv[2], sys.argv[3]
try:
    rows = db.execute("select kernel_name, avg(value), count(*) from counters_collection where counter_name=? group by kernel_name", (c,)).fetchall()
except Exception as e:
    rows = []; print("ERROR", b, c, e)
dur = {r[0]: r[1] for r in db.execute("select name, avg(end-start) from kernels group by name").fetchall()}
for name, v, n in rows:
    if "fps" in name:
        print("%s\t%s\t%s\t%.6g\t%d\t%.4f\t%s" % (b, c, name.split("(")[0][:60], v, n, dur.get(name, 0) / 1e6, sys.argv[4]))
PY
  done
done
done
python - $OUT <<'PY'
import json, sys
sys.path.insert(0, ".")
import bench
out = sys.argv[1]
res = {"_source_blobs": {f: bench.git_blob_sha1(f) for f in bench.FPS_KERNEL_SOURCES},     # bench.py refuses the counts on other sources
       "_note": "rocprofv3 --pmc, one counter per pass, bench.py --workload c2 --kind K --batch B -> key bB_K; values per launch (average over the traced launches); "
                "sq_insts_valu = wave64 VALU instructions; physical VALU share = sq_insts_valu * 64 / duration / (1024 * 32 * 2.4e9)"}
for ln in open(out + "/fps_valu_rows.txt"):
    p = ln.rstrip("\n").split("\t")
    if len(p) != 7:
        continue
    b, c, name, v, n, ms, kind = p
    e = res.setdefault("b%s_%s" % (b, kind), {"kernel": name})
    e[c.lower() + "_per_launch"] = float(v)
    e["traced_ms"] = float(ms)
for k, e in res.items():
    if isinstance(e, dict) and "sq_insts_valu_per_launch" in e:
        e["valu_share_of_chip_traced"] = e["sq_insts_valu_per_launch"] * 64 / (e["traced_ms"] * 1e-3) / (1024 * 32 * 2.4e9)
json.dump(res, open(out + "/traffic_fps_valu.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
