#!/bin/bash
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
O=gpurun_out/r04c29; mkdir -p $O
timeout 900 python -m pytest tests -x -q -m gpu -k "train or sampling_plan or consecutive or recycled or backward or bn_relu or wgrad" > $O/pytest_train.log 2>&1; echo "train rc $?"; tail -3 $O/pytest_train.log
for flag in False True False True; do
python - <<PY 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); print('CONV1X1_TRAIN_AS_GEMM=$flag', round(d['value'],1), d['unit'], '%.2f ms per step' % d['ms_per_step'])"
import sys, runpy
import ws3d_amd.nn_blocks as nb
nb.CONV1X1_TRAIN_AS_GEMM = $flag
sys.argv = ['bench.py', '--workload', 't1', '--no-cpu-baseline']
runpy.run_path('bench.py', run_name='__main__')
PY
done
