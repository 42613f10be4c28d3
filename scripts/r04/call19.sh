#!/bin/bash
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
O=gpurun_out/r04c19; mkdir -p $O
timeout 900 bash scripts/ubench/sa1_compact_ablation.sh > $O/sa1_compact_ablation.txt 2>&1; echo "rc $?"; grep -v amdgpu.ids $O/sa1_compact_ablation.txt | tail -8
