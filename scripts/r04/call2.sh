#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r04c2; mkdir -p $OUT; export TMPDIR=/tmp
T="timeout 900"
for m in train eval eval_nofast; do $T python scripts/poison_forward.py $m; done > $OUT/poison_forward.txt 2>&1
grep "RESULT\|DIFF" $OUT/poison_forward.txt | head -40
$T python bench.py --workload c5 2>/dev/null | tail -1 > $OUT/bench_c5_b8.json
$T python bench.py 2>$OUT/bench_default.err | tail -1 > $OUT/bench_default.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04c2/bench_default.json"))
print("value",d["value"],"lat",d["latency_mode"]["ms_per_batch"])
for k in d["kernels"]: print("%-90s %.4f ms  launches %.1f traffic %s" % (k["name"][:90],k["ms_per_step"],k["launches_per_step"], (k.get("traffic_bytes_per_launch") or {}).get("hbm_bytes")))
PY
