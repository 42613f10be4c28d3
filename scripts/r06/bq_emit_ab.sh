#!/bin/bash
# round 6: the fused query + group kernel after the emit rework -- parity tests of the search / grouping operators, then the c2 block
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
out=gpurun_out/bq_emit
mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_independent_search.py -x -q -m gpu -k "ball_query or query_and_group or group or sa1 or c1_gpu or search" > $out/pytest.log 2>&1; tail -3 $out/pytest.log
timeout 300 python scripts/ab_bq.py > $out/ab_bq.txt 2>&1; cat $out/ab_bq.txt
timeout 300 python bench.py --full-line --workload c2 --batch 512 --no-cpu-baseline 2>$out/c2.err | tail -1 > $out/bench_c2_b512.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/bq_emit/bench_c2_b512.json"))
print({k:d.get(k) for k in ("value","ms_per_step")})
for k in d.get("kernels",[]): print(k.get("name"), k.get("ms_per_step"), k.get("frac_of_8TBps"))
PY
