#!/bin/bash
# Run on the GPU box (through gpurun): full GPU test suite, the bench lines and the rocprofv3
# evidence of one code state.  Everything lands in gpurun_out/refresh/; copy what should be
# judged into profiles/ afterwards (see profiles/README.md).  The default bench line is the BASELINE headline (c3 + the c2 block)
# on the hdl64 generator (round 3); `--kind lidar` is the sparse generator of rounds 1-2.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/refresh
mkdir -p $OUT
export TMPDIR=/tmp
T="timeout 900"
$T python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; tail -3 $OUT/pytest_gpu.log
# (the default lines are taken LAST, after this run's counter passes have been copied into profiles/: they quote them)
$T python bench.py --full-line --steps 160 --no-cpu-baseline --no-side-runs --c2-batch 0 2>>$OUT/bench_default.err | tail -1 > $OUT/bench_c3_b8_steps160.json
$T python bench.py --full-line --pipeline-depth 1 --no-cpu-baseline --no-side-runs --c2-batch 0 2>>$OUT/bench_default.err | tail -1 > $OUT/bench_c3_b8_depth1.json
GPU_MAX_HW_QUEUES=4 $T python bench.py --full-line --pipeline-depth 3 --no-cpu-baseline --no-side-runs --c2-batch 0 2>>$OUT/bench_default.err | tail -1 > $OUT/bench_c3_b8_depth3_queues4.json
$T python bench.py --full-line --workload c2                              2>$OUT/bench_c2.err | tail -1 > $OUT/bench_c2_b512.json
for b in 1 8 64 256 1024; do
  $T python bench.py --full-line --workload c2 --batch $b --no-cpu-baseline 2>>$OUT/bench_c2.err | tail -1 > $OUT/bench_c2_b$b.json
done
$T python bench.py --full-line --workload c5               2>$OUT/bench_c5_b8.err   | tail -1 > $OUT/bench_c5_b8.json
$T python bench.py --full-line --workload s2               2>$OUT/bench_s2_b800.err | tail -1 > $OUT/bench_s2_b800.json
$T python bench.py --full-line --workload t1 --no-cpu-baseline 2>$OUT/bench_t1_b8.err | tail -1 > $OUT/bench_t1_b8.json
for w in c2 c3 c5 s2; do
  extra=""; [ $w = c3 ] && extra="--pipeline-depth 1 --no-graph --c2-batch 0"
  rm -rf /tmp/prof_$w
  $T rocprofv3 --kernel-trace --stats -d /tmp/prof_$w -o $w -- python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline $extra > $OUT/prof_$w.log 2>&1
  db=$(find /tmp/prof_$w -name '*.db' | head -1)
  python scripts/rocpd_stats.py "$db" > $OUT/${w}_kernel_stats.csv 2>>$OUT/prof_$w.log
done
# the default command itself (what the driver runs), kernel trace only
rm -rf /tmp/prof_default
$T rocprofv3 --kernel-trace --stats -d /tmp/prof_default -o d -- python bench.py --no-cpu-baseline > $OUT/prof_default.log 2>&1
python scripts/rocpd_stats.py "$(find /tmp/prof_default -name '*.db' | head -1)" > $OUT/default_kernel_stats.csv 2>>$OUT/prof_default.log
# HBM traffic: separate --pmc passes (MI355X_MICROARCH.md), c2 at the default batch 512 and c5
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  $T rocprofv3 --pmc $c --kernel-trace -d /tmp/pmc_$c -o pmc -- python bench.py --workload c2 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmc_$c.log 2>&1
done
python scripts/pmc_traffic.py "$(find /tmp/pmc_FETCH_SIZE -name '*.db' | head -1)" "$(find /tmp/pmc_WRITE_SIZE -name '*.db' | head -1)" \
  $OUT/traffic.json "c2 batch 512, bytes per launch, rocprofv3 --pmc in separate passes" 512 > /dev/null 2>>$OUT/pmc_WRITE_SIZE.log
for wl in c5:8; do
  w=${wl%%:*}; nb=${wl##*:}
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_${w}_$c
    $T rocprofv3 --pmc $c --kernel-trace -d /tmp/pmc_${w}_$c -o pmc -- python bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmc_${w}_$c.log 2>&1
  done
  python scripts/pmc_traffic.py "$(find /tmp/pmc_${w}_FETCH_SIZE -name '*.db' | head -1)" "$(find /tmp/pmc_${w}_WRITE_SIZE -name '*.db' | head -1)" \
    $OUT/traffic_$w.json "$w batch $nb, bytes per launch, rocprofv3 --pmc in separate passes" $nb > /dev/null 2>>$OUT/pmc_${w}_WRITE_SIZE.log
done
# round-2 extras: FPS kernels side by side, the eager step's kernel timeline (side streams), roipool3d ablation, ubenches
{ python scripts/ab_fps.py default; WS3D_FPS_BUCKET=0 python scripts/ab_fps.py dense 8x16384x4096 256x16384x4096 512x16384x4096 8x12345x3000; WS3D_FPS_ROUNDS=0 python scripts/ab_fps.py one-sample-per-exchange 8x16384x4096 256x16384x4096 512x16384x4096; } > $OUT/fps_ab.txt 2>/dev/null
# HBM traffic of the eager c3 step per launch family (separate --pmc passes) -> traffic_c3.json
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_c3_$c
  $T rocprofv3 --pmc $c --kernel-trace -d /tmp/pmc_c3_$c -o pmc -- python bench.py --workload c3 --pipeline-depth 1 --no-graph --c2-batch 0 --steps 4 --warmup 2 --no-cpu-baseline --no-side-runs > $OUT/pmc_c3_$c.log 2>&1
done
python scripts/pmc_traffic_c3.py "$(find /tmp/pmc_c3_FETCH_SIZE -name '*.db' | head -1)" "$(find /tmp/pmc_c3_WRITE_SIZE -name '*.db' | head -1)" \
  $OUT/traffic_c3.json hdl64 8 > $OUT/traffic_c3_summary.txt 2>>$OUT/pmc_c3_WRITE_SIZE.log
$T bash scripts/pmc_fps_valu.sh $OUT/pmc_fps > $OUT/pmc_fps_valu.txt 2>&1
$T bash scripts/ubench/fps_rounds2_prof.sh > $OUT/fps_rounds2_segments.txt 2>&1
$T python scripts/graph_fork_debug.py > $OUT/graph_fork_join_stress.txt 2>&1
$T python scripts/ubench/compact_vs_dense.py > $OUT/compact_vs_dense_dispatch.txt 2>&1
rm -rf /tmp/tl; (cd /tmp && $T rocprofv3 --kernel-trace -d /tmp/tl -o t -- python $OLDPWD/scripts/host_issue_time.py > $OLDPWD/$OUT/host_issue_time.txt 2>&1)
python scripts/rocpd_timeline.py "$(find /tmp/tl -name '*.db' | head -1)" fps_rounds2_kernel $OUT/c3_eager_timeline.txt > /dev/null 2>>$OUT/host_issue_time.txt
python scripts/host_issue_time.py > $OUT/host_issue_time_untraced.txt 2>&1
$T bash scripts/ablate_roi.sh > $OUT/roipool3d_ablation.txt 2>&1
(cd scripts/ubench && hipcc -O3 --offload-arch=gfx950 row_copy.hip -o /tmp/row_copy 2>/dev/null && timeout 120 /tmp/row_copy) > $OUT/ubench_row_copy.txt 2>&1
$T python scripts/exp_latency_segments.py 30 > $OUT/latency_segments.txt 2>&1
$T bash scripts/ubench/sa1_compact_ablation.sh > $OUT/sa1_compact_ablation.txt 2>&1
timeout 1500 bash scripts/throughput_marginal.sh hdl64 > $OUT/throughput_marginal_cost.txt 2>&1
# the default lines, quoting THIS run's counter passes (profiles/ of the box's scratch copy)
cp $OUT/traffic.json $OUT/traffic_c5.json $OUT/traffic_c3.json profiles/ 2>/dev/null
[ -s $OUT/pmc_fps/traffic_fps_valu.json ] && cp $OUT/pmc_fps/traffic_fps_valu.json profiles/
$T python bench.py --full-line                                            2>$OUT/bench_default.err | tail -1 > $OUT/bench_default.json
$T python bench.py --full-line --kind lidar                               2>$OUT/bench_default_lidar.err | tail -1 > $OUT/bench_default_lidar.json
$T python bench.py --full-line --steps 20 --warmup 5                      2>>$OUT/bench_default.err | tail -1 > $OUT/bench_default_steps20_warmup5.json
ls -la $OUT
