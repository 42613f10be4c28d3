#!/bin/bash
# round 6: ws3d_tune key 5 (search launches on 8 waves per tile in the pipeline's throughput geometry): the new test, the GPU suite, the 20-deep line with and without it
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; out=gpurun_out/bq_tune; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "eight_waves" 2>&1 | tail -2
line() { python bench.py --full-line --no-cpu-baseline --no-side-runs --c2-batch 0 --steps 160 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1: %.1f scenes/s  %.4f ms/step' % (d['value'], d['ms_per_step']))"; }
line "throughput geometry incl. key 5" | tee $out/ab.txt
WS3D_NO_BQ_TUNE=1 python - <<'PY' | tee -a $out/ab.txt
import subprocess, sys, json, os
import ws3d_amd.pipeline as p
# the same line with key 5 left at its default: patch the table in a child process
code = "import sys; sys.argv=['bench.py','--full-line','--no-cpu-baseline','--no-side-runs','--c2-batch','0','--steps','160']; import ws3d_amd.pipeline as p; p.THROUGHPUT_GEOMETRY.pop('bq_wide_nw'); import runpy; runpy.run_path('bench.py', run_name='__main__')"
r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
d = json.loads(r.stdout.strip().splitlines()[-1]); print("without key 5: %.1f scenes/s  %.4f ms/step" % (d["value"], d["ms_per_step"]))
PY
line "throughput geometry incl. key 5 (again)" | tee -a $out/ab.txt
timeout 1500 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1; tail -2 $out/pytest_gpu.log
