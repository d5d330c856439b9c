#!/bin/bash
# Run on the GPU box (through gpurun): the end-of-round evidence in one call -- the GPU test suite, smoke(), the bench with its
# rocprofv3 kernel statistics and PMC passes, the per-op table, the other BASELINE configurations, the train and FPS benches.
# Usage: tools/final_round.sh [tag]   -> gpurun_out/<tag>/..., gpurun_out/<tag>_pmc/pmc_summary.json  (copy what is kept to profiles/)
TAG=${1:-r06_a}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$TAG
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -5 > gpurun_out/$TAG/pytest_gpu.txt
cat gpurun_out/$TAG/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
bash tools/prof_bench.sh $TAG > gpurun_out/$TAG/prof.log 2>&1; tail -2 gpurun_out/$TAG/prof.log | cut -c1-600
bash tools/pmc_traffic.sh ${TAG}_pmc > gpurun_out/$TAG/pmc.log 2>&1; tail -c 600 gpurun_out/$TAG/pmc.log
timeout 900 python tools/ops_bench.py > gpurun_out/$TAG/ops_microbench.json 2> gpurun_out/$TAG/ops.log; tail -c 300 gpurun_out/$TAG/ops_microbench.json
timeout 600 python tools/config_bench.py > gpurun_out/$TAG/configs.json 2>> gpurun_out/$TAG/ops.log
timeout 300 python tools/train_bench.py --warmup 30 > gpurun_out/$TAG/train_b8_bench.json 2>> gpurun_out/$TAG/ops.log
timeout 300 python tools/train_bench.py --batch 32 > gpurun_out/$TAG/train_b32_bench.json 2>> gpurun_out/$TAG/ops.log
bash tools/prof_train.sh ${TAG}_train 8 > gpurun_out/$TAG/prof_train.log 2>&1
timeout 300 python tools/emd_bench.py > gpurun_out/$TAG/emd_bench.txt 2>> gpurun_out/$TAG/ops.log
timeout 300 python tools/train_bench.py --dtype bf16 --warmup 30 > gpurun_out/$TAG/train_b8_bf16_bench.json 2>> gpurun_out/$TAG/ops.log
timeout 300 python tools/fps_bench.py > gpurun_out/$TAG/fps_bench.txt 2>> gpurun_out/$TAG/ops.log
for b in 32 64; do timeout 300 python tools/train_bench.py --dtype bf16 --batch $b > gpurun_out/$TAG/train_b${b}_bf16_bench.json 2>> gpurun_out/$TAG/ops.log; done
timeout 300 python tools/train_bench.py --batch 64 > gpurun_out/$TAG/train_b64_f32_bench.json 2>> gpurun_out/$TAG/ops.log
timeout 300 python tools/debug/bf16_gemm_bench.py > gpurun_out/$TAG/bf16_gemm_bench.txt 2>&1
timeout 300 python tools/debug/bf16_tn_bench.py > gpurun_out/$TAG/bf16_tn_bench.txt 2>&1
bash tools/trace_train.sh ${TAG}_trace 8 f32 > gpurun_out/$TAG/trace.log 2>&1
bash tools/trace_train.sh ${TAG}_trace_bf16 8 bf16 > gpurun_out/$TAG/trace_bf16.log 2>&1
bash tools/prof_train.sh ${TAG}_train_bf16 8 "--dtype bf16" > gpurun_out/$TAG/prof_train_bf16.log 2>&1
# round 6: lab evidence (per-phase stamps of a dense block and of the xyz k-NN, the split-bf16 GEMM's two kernels and its operand-stream
# switches, what the pipelined all-gather costs on a one-rank RCCL group)
timeout 300 python tools/debug/stem_stamps.py > gpurun_out/$TAG/stem_stamps.txt 2>&1
timeout 300 python tools/debug/knn_stamps.py > gpurun_out/$TAG/knn_stamps.txt 2>&1
( timeout 300 python tools/debug/x3_lab.py; for f in "-DS3_NOW" "-DS3_NOX" "-DS3_NOW -DS3_NOX" "-DS3_NOSCHED"; do timeout 300 python tools/debug/x3_lab.py $f; done ) 2>&1 | grep "us per launch\|bit-identical" > gpurun_out/$TAG/x3_lab.txt
timeout 300 python tools/debug/gather_cost.py 2>&1 | grep "ms per step" > gpurun_out/$TAG/gather_cost.txt
echo done
