#!/bin/bash
# The launches of ONE graph replay of the c3 step, by kernel: rocprofv3 kernel traces of bench.py at two step counts, the difference
# of the call counts divided by the difference of the steps (warm-up, capture and the latency-mode runs cancel).
#   bash scripts/launches_per_step.sh [kind] > gpurun_out/launches_per_step.txt
cd "$(dirname "$0")/.."
KIND=${1:-hdl64}
export TMPDIR=/tmp WS3D_TUNE_GEMMS=0
for S in 20 60; do
  rm -rf /tmp/prof_lps_$S
  timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_lps_$S -o l -- python bench.py --kind $KIND --steps $S --warmup 2 --no-side-runs --no-cpu-baseline --c2-batch 0 > /dev/null 2>&1
  python scripts/rocpd_stats.py $(find /tmp/prof_lps_$S -name "*.db" | head -1) > /tmp/lps_$S.csv
done
python - <<'P'
import csv
a = {r["Name"]: r for r in csv.DictReader(open("/tmp/lps_20.csv"))}
b = {r["Name"]: r for r in csv.DictReader(open("/tmp/lps_60.csv"))}
rows, tot = [], 0.0
for name, r in b.items():
    d = (int(r["Calls"]) - int(a.get(name, {"Calls": 0})["Calls"])) / 40.0
    if d > 0:
        rows.append((d, float(r["AverageNs"]) / 1e3, name)); tot += d
for d, us, name in sorted(rows, key=lambda t: -t[0] * t[1]):
    print("%5.2f x %8.1f us  %s" % (d, us, name[:150]))
print("launches per step: %.2f" % tot)
P
