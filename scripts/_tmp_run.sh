timeout 600 python -m pytest tests/test_gpu_parity.py -q -k "mlp2_rows" 2>&1 | tail -2
timeout 600 python -m pytest tests/test_golden.py tests/test_stage1.py -q -m gpu 2>&1 | tail -2
for v in 1 0 1 0; do WS3D_FUSED_MLP2_ROWS=$v timeout 900 python bench.py --no-cpu-baseline --c2-batch 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('MLP2=$v', round(d['value']), d['ms_per_step'], d['latency_mode']['ms_per_batch'])"; done
