cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; R=$PWD
rm -rf /tmp/tp /tmp/eg
(cd /tmp && timeout 600 rocprofv3 --kernel-trace -d /tmp/tp -o tp -- python $R/bench.py --no-cpu-baseline --c2-batch 0 --steps 120 > $R/gpurun_out/conc_tp.log 2>&1)
(cd /tmp && timeout 600 rocprofv3 --kernel-trace -d /tmp/eg -o eg -- python $R/bench.py --no-cpu-baseline --c2-batch 0 --steps 40 --pipeline-depth 1 --no-graph > $R/gpurun_out/conc_eg.log 2>&1)
tail -1 gpurun_out/conc_tp.log | cut -c1-300
python scripts/rocpd_concurrency.py $(find /tmp/tp -name '*.db' | head -1) $(find /tmp/eg -name '*.db' | head -1) 100
