#!/bin/bash
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
O=gpurun_out/r04c40; mkdir -p $O
timeout 1500 python scripts/exp_fastpath_ab.py FUSED_COMPACT3_MAX_LDS 65536 163840 160 > $O/exp_compact3_lds.txt 2>&1; grep -v amdgpu.ids $O/exp_compact3_lds.txt | tail -10
