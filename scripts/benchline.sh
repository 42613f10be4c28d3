python bench.py --full-line --no-side-runs --no-cpu-baseline --c2-batch 0 "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1])
print('value %.0f  ms/batch %.4f  latency %.3f ms' % (d['value'], d['ms_per_step'], d['latency_mode']['ms_per_batch']))
for k in d['kernels']: print('  %-70s %.4f' % (k['name'][:70], k['ms_per_step']))"
