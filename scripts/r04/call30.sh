#!/bin/bash
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
O=gpurun_out/r04c30; mkdir -p $O
timeout 900 python scripts/exp_miopen_conv_bits.py 2000 > $O/miopen_conv_bits.txt 2>&1; grep -v amdgpu.ids $O/miopen_conv_bits.txt | tail -20
