cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out/chk
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/chk/pytest_gpu.log 2>&1; tail -3 gpurun_out/chk/pytest_gpu.log
timeout 600 python bench.py --steps 20 --warmup 5 --detail gpurun_out/chk/detail.json 2>gpurun_out/chk/bench.err > gpurun_out/chk/line.json; python -c "
import json; d=json.load(open('gpurun_out/chk/line.json')); print({k:d.get(k) for k in ('value','ms_per_step','value_steady','latency_ms','value_lidar','c2_scenes_per_s','c2_query_group_ms','c2_query_group_hbm_frac')})"
