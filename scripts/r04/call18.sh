#!/bin/bash
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
O=gpurun_out/r04c18; mkdir -p $O
timeout 600 python scripts/exp_latency_segments.py 30 > $O/exp_latency_segments.txt 2>&1; echo "rc $?"; grep -v amdgpu.ids $O/exp_latency_segments.txt | tail -8
