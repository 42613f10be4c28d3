#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/nested; mkdir -p $OUT
python -m ws3d_amd.build > /dev/null
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_golden.py -x -q -m gpu -k "fps or nested or sampling or stage1_forward or kernel_variants" > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
python scripts/r06/nested_verdicts.py 2>&1 | grep -v amdgpu | grep -E "False|scenes" | tee $OUT/nested_verdicts.txt
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --c2-batch 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('driver flags', d['value'], 'steady', d.get('value_steady'), 'latency', d.get('latency_ms'))" | tee $OUT/line.txt
