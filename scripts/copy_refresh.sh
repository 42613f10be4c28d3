#!/bin/bash
# gpurun_out/refresh (scripts/gpu_refresh.sh) -> profiles/, under the names profiles/README.md lists (round prefix as $1, default r03)
cd "$(dirname "$0")/.."
R=${1:-r04}; S=gpurun_out/refresh; D=profiles
for f in bench_default.json bench_default_lidar.json bench_c3_b8_steps160.json bench_c3_b8_depth1.json bench_c3_b8_depth3_queues4.json \
         bench_c2_b1.json bench_c2_b8.json bench_c2_b64.json bench_c2_b256.json bench_c2_b512.json bench_c2_b1024.json \
         bench_c5_b8.json bench_s2_b800.json bench_t1_b8.json \
         c2_kernel_stats.csv c3_kernel_stats.csv c5_kernel_stats.csv s2_kernel_stats.csv default_kernel_stats.csv \
         fps_ab.txt fps_rounds2_segments.txt graph_fork_join_stress.txt compact_vs_dense_dispatch.txt \
         c3_eager_timeline.txt roipool3d_ablation.txt pytest_gpu.log bench_default_steps20_warmup5.json \
         throughput_marginal_cost.txt sa1_compact_ablation.txt launches_per_step.txt bq_emit_instr_per_byte.txt traffic_c3_summary.txt \
         bench_ops_b8.json bench_ops_b256.json bench_ops_b8_detail.json bench_ops_b256_detail.json bench_default_line.json bench_default_detail.json bench_default_steps20_warmup5_detail.json; do
  [ -s $S/$f ] && cp $S/$f $D/${R}_$f
done
[ -s $S/host_issue_time_untraced.txt ] && cp $S/host_issue_time_untraced.txt $D/${R}_host_issue_time.txt
for f in traffic.json traffic_c5.json traffic_c3.json traffic_ops.json traffic_ops256.json; do [ -s $S/$f ] && cp $S/$f $D/$f; done
[ -s $S/pmc_fps/traffic_fps_valu.json ] && cp $S/pmc_fps/traffic_fps_valu.json $D/traffic_fps_valu.json
[ -s $S/pmc_fps_valu.txt ] && cp $S/pmc_fps_valu.txt $D/${R}_fps_valu_counters.txt
ls $D | wc -l
