#!/bin/bash
# Round 6: why do the own fp32-MFMA kernels run at 0.26 of the matrix peak?  SQ counters per kernel of the eager c3 step
# (rocprofv3 --pmc serialises the dispatches: isolated figures), separate passes (8 SQ slots per pass), kernel trace only.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=${1:-gpurun_out/pmc_mfma}; mkdir -p $OUT; export TMPDIR=/tmp
REPO=$PWD
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU"
P2="SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_MFMA"
P3="GRBM_GUI_ACTIVE SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_BUSY_CU_CYCLES"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1)); rm -rf /tmp/pm_$i
  (cd /tmp && timeout 400 rocprofv3 --pmc $P --kernel-trace -d /tmp/pm_$i -o pm -- python $REPO/bench.py --workload c3 --pipeline-depth 1 --no-graph --c2-batch 0 --steps 3 --warmup 2 --no-cpu-baseline --no-side-runs > $REPO/$OUT/log_$i.txt 2>&1)
  python $REPO/scripts/r06/pmc_mfma_sum.py "$(find /tmp/pm_$i -name '*.db' | head -1)" >> $OUT/counters.txt 2>&1
done
cat $OUT/counters.txt
