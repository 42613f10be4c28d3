#!/bin/bash
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); print(round(d['value']), d['unit'], d['ms_per_step'], d['roofline']['frac'], d['cpu_baseline']['value'])"
