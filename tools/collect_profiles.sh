#!/bin/bash
# Copy the evidence of one tools/final_round.sh + tools/pmc_valu_ops.sh run from gpurun_out/ (scratch) into profiles/ (tracked).
# Usage: tools/collect_profiles.sh <tag>      e.g. r04_e
set -e
T=$1
G=gpurun_out
P=profiles
for f in bench.json configs.json emd_bench.txt fps_bench.txt kernel_stats.csv ops_microbench.json pytest_gpu.txt train_b32_bench.json \
         train_b8_bench.json train_b8_bf16_bench.json; do
    cp $G/$T/$f $P/${T}_$f
done
cp $G/$T/bench_profiled.json $P/${T}_bench_under_rocprof.json
python3 tools/stamp_profile.py $G/${T}_pmc/pmc_summary.json $P/${T}_pmc_summary.json --latest    # + commit stamp, refreshes pmc_traffic_latest.json
cp $G/${T}_train/kernel_stats.csv $P/${T}_train_b8_kernel_stats.csv
cp $G/${T}_trace/timeline.txt $P/${T}_train_timeline.txt
for f in train_b32_bf16_bench.json train_b64_bf16_bench.json train_b64_f32_bench.json bf16_gemm_bench.txt bf16_tn_bench.txt; do [ -f $G/$T/$f ] && cp $G/$T/$f $P/${T}_$f; done
for f in stem_stamps.txt knn_stamps.txt x3_lab.txt gather_cost.txt; do [ -f $G/$T/$f ] && cp $G/$T/$f $P/${T}_$f; done
[ -f $G/${T}_trace_bf16/timeline.txt ] && cp $G/${T}_trace_bf16/timeline.txt $P/${T}_train_bf16_timeline.txt
[ -f $G/${T}_train_bf16/kernel_stats.csv ] && cp $G/${T}_train_bf16/kernel_stats.csv $P/${T}_train_b8_bf16_kernel_stats.csv
for f in rccl_world1_dry_run.json rccl_world1_bench.json gather_overlap_dry_run.json; do [ -f $G/$f ] && cp $G/$f $P/${T}_$f; done
[ -f $G/${T}_valu/valu_summary.json ] && cp $G/${T}_valu/valu_summary.json $P/${T}_valu_summary.json
[ -f $G/bench_side_tables.json ] && cp $G/bench_side_tables.json $P/bench_side_tables.json
ls $P | grep ${T}_ | wc -l
