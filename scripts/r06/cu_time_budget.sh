cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/cu
rm -rf /tmp/ct
timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/ct -o ct -- python scripts/r06/cu_time_budget.py run > gpurun_out/cu/run.log 2>&1
f=$(find /tmp/ct -name '*kernel_trace.csv' | head -1)
python scripts/r06/cu_time_budget.py report $f > gpurun_out/cu/cu_time_budget.txt 2>&1
tail -60 gpurun_out/cu/cu_time_budget.txt
