#!/bin/bash
# what the fused query + group kernel of the c2 block issues per 256 bytes it stores -- SQ counters of
# ball_query_grid_coop_kernel<fused> at 512 scenes (rocprofv3 --pmc, one pass, kernel trace only), against its duration.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=${1:-gpurun_out}; export TMPDIR=/tmp
rm -rf /tmp/pmc_bqe
(cd /tmp && timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --kernel-trace -d /tmp/pmc_bqe -o p -- python $OLDPWD/bench.py --workload c2 --batch 512 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1)
python - "$(find /tmp/pmc_bqe -name '*.db' | head -1)" <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select kernel_name, counter_name, avg(value) from counters_collection group by kernel_name, counter_name").fetchall()
dur = {r[0]: r[1] for r in db.execute("select name, avg(end-start) from kernels group by name").fetchall()}
per = {}
for k, c, v in rows:
    per.setdefault(k, {})[c] = v
B, M, NS, C = 512, 4096, 64, 1
stored = B * ((3 + C) * M * NS + M * NS) * 4.0          # grouped tensor + lists, bytes per launch
for k, d in per.items():
    if "ball_query_grid_coop_kernel" not in k:
        continue
    print("# %s\n# %d scenes per launch, %.3f ms traced (counter pass: serialised dispatches), %.3f GB stored per launch" % (k[:110], B, dur.get(k, 0) / 1e6, stored / 1e9))
    u = stored / 256.0
    for c in sorted(d):
        print("%-20s %14.0f per launch   %8.2f per 256 B stored (wave64 instructions: x 64 lanes)" % (c, d[c], d[c] / u))
    issue = (d.get("SQ_INSTS_VALU", 0) + d.get("SQ_INSTS_SALU", 0) + d.get("SQ_INSTS_LDS", 0) + d.get("SQ_INSTS_VMEM_RD", 0) + d.get("SQ_INSTS_VMEM_WR", 0))
    print("instructions issued per launch %.3g; per CU and microsecond of the traced duration: %.1f" % (issue, issue / 256.0 / (dur.get(k, 1) / 1e3)))
PY
