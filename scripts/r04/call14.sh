#!/bin/bash
# round 4, call 14: the default line with throughput_mode.matrix_roofline, the contract test that checks it, marginal costs per family
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
O=gpurun_out/r04c14; mkdir -p $O
timeout 900 python -m pytest tests/test_bench_contract.py -x -q -m gpu -k "default_line or follows_the_contract" > $O/pytest_contract.log 2>&1; echo "contract rc $?" 
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?"
timeout 1500 bash scripts/throughput_marginal.sh hdl64 > $O/throughput_marginal_cost.txt 2>&1; echo "marginal rc $?"
tail -3 $O/pytest_contract.log
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04c14/bench_default.json').read().strip().splitlines()[-1])
print(d['value'], d['throughput_mode']['matrix_roofline'])
PY
cat $O/throughput_marginal_cost.txt
