#!/bin/bash
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
O=gpurun_out/r04c33; mkdir -p $O
timeout 600 python scripts/ubench/three_nn_query_order.py > $O/three_nn_query_order.txt 2>&1; grep -v amdgpu.ids $O/three_nn_query_order.txt | tail -12
