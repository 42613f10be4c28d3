#!/bin/bash
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
O=gpurun_out/r04c26; mkdir -p $O
timeout 1200 python scripts/exp_fps_priority.py 80 > $O/exp_fps_priority.txt 2>&1; echo "rc $?"; grep -v amdgpu.ids $O/exp_fps_priority.txt | tail -12
