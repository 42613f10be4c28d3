#!/bin/bash
# rocprofv3 kernel trace of the Stage-1 training loop (synthetic scenes, batch 8); the summary
# covers the last 1.2 s of the trace only (steady state, after MIOpen's solver search)
cd ${GRAFT_REPO_ROOT:-/root/repo}; export TMPDIR=/tmp; mkdir -p gpurun_out/train
rm -rf /tmp/prof_tr
rocprofv3 --kernel-trace -d /tmp/prof_tr -o tr -- python -m ws3d_amd.train_rpn --synthetic 32 --batch_size 8 --total_iters 60 --ckpt_save_interval 1000 --output_dir /tmp/tr_out > gpurun_out/train/log.txt 2>&1
db=$(find /tmp/prof_tr -name '*.db' | head -1)
python scripts/rocpd_stats.py "$db" gpurun_out/train/train_kernel_stats.csv --last-ms 1200 > /dev/null
python - <<'PY'
import csv
rows = list(csv.DictReader(open("gpurun_out/train/train_kernel_stats.csv")))
steps = [int(r["Calls"]) for r in rows if "fps_reg_kernel<32" in r["Name"]] or [1]
print("steps in window:", steps[0], " total kernel ms/step: %.2f" % (sum(int(r["TotalDurationNs"]) for r in rows) / steps[0] / 1e6))
for r in rows[:40]:
    print("%8.3f ms/step %6.1f calls/step  %s" % (int(r["TotalDurationNs"]) / steps[0] / 1e6, int(r["Calls"]) / steps[0], r["Name"][:110]))
PY
grep -v "^[WE]2026" gpurun_out/train/log.txt | tail -4
python - "$db" <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
print(cols)
g = [c for c in cols if "grid" in c.lower()][:3]
q = "select name, %s, end-start from kernels where name like '%%det_wave%%' order by start desc limit 12" % ", ".join(g)
for r in db.execute(q): print(r[0][:40], r[1:])
PY
