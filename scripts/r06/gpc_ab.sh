#!/bin/bash
# round 6: the compact last-layer kernel (gemm_pool_compact_pair_kernel) with four operand tiles in flight against one: parity tests, the kernel's
# duration in the eager c3 step (rocprofv3 kernel trace) and the 20-deep line, per build
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
out=gpurun_out/gpc
mkdir -p $out
: > $out/ab.txt
[ "${GPC_TESTS:-1}" = 1 ] && { timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_golden.py -x -q -m gpu -k "compact or fast_path or sa_fp or stage1 or headline or gemm_pool or chain" > $out/pytest.log 2>&1; tail -3 $out/pytest.log | tee -a $out/ab.txt; }
run() {
  echo "== $1" | tee -a $out/ab.txt
  rm -rf /tmp/gpc_prof
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/gpc_prof -o p -- python bench.py --workload c3 --steps 10 --warmup 3 --no-cpu-baseline --pipeline-depth 1 --no-graph --c2-batch 0 --no-side-runs > $out/prof.log 2>&1
  python scripts/rocpd_stats.py "$(find /tmp/gpc_prof -name '*.db' | head -1)" 2>/dev/null | python -c "
import csv, sys
for r in csv.reader(sys.stdin):
    if len(r) > 3 and any(k in r[0] for k in ('gemm_pool_compact', 'pgather_gemm2_compact', 'interp_gemm_big', 'chain_mlp3')):
        print('  %-60s calls %5s  avg %9.1f ns' % (r[0][:60], r[1], float(r[3])))" | tee -a $out/ab.txt
  python bench.py --full-line --no-cpu-baseline --no-side-runs --c2-batch 0 --steps 160 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c3 160 steps: %.1f scenes/s  %.4f ms/step' % (d['value'], d['ms_per_step']))" | tee -a $out/ab.txt
}
run default
for v in "$@"; do
  WS3D_EXTRA_DEFS="$v" python -m ws3d_amd.build --only gemm_pool.hip > /dev/null 2>$out/build.err || { echo "build failed: $v"; tail -5 $out/build.err; continue; }
  run "$v"
done
python -m ws3d_amd.build --only gemm_pool.hip > /dev/null
run "default again"
