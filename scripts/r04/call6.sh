#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r04c6; mkdir -p $OUT; export TMPDIR=/tmp
T="timeout 1200"
$T python bench.py 2>$OUT/bench_default.err | tail -1 > $OUT/bench_default.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04c6/bench_default.json"))
print("value",d["value"],"lat",d["latency_mode"]["ms_per_batch"], d["latency_mode"]["rank0_ms_by_launch"], "lidar", d["other_generator"]["value"], d["other_generator"]["latency_ms_per_batch"])
print("c2", d["c2"]["scenes_per_s_per_gpu"], [ (k["name"][:30], k["ms_per_step"]) for k in d["c2"]["kernels"]])
for k in d["kernels"]: print("%-90s %.4f ms  launches %.1f" % (k["name"][:90],k["ms_per_step"],k["launches_per_step"]))
PY
$T python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; tail -5 $OUT/pytest_gpu.log
