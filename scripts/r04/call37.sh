#!/bin/bash
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
O=gpurun_out/r04c37; mkdir -p $O
timeout 600 python scripts/glue_ops.py > $O/glue_ops.txt 2>&1; grep -v amdgpu.ids $O/glue_ops.txt | cut -c1-260 | tail -40
