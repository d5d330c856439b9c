cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02_d
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -5 > gpurun_out/r02_d/pytest_gpu.txt
cat gpurun_out/r02_d/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
bash tools/prof_bench.sh r02_d > gpurun_out/r02_d/prof.log 2>&1; tail -2 gpurun_out/r02_d/prof.log | cut -c1-600
bash tools/pmc_traffic.sh r02_d_pmc > gpurun_out/r02_d/pmc.log 2>&1; tail -c 600 gpurun_out/r02_d/pmc.log
timeout 900 python tools/ops_bench.py > gpurun_out/r02_d/ops_microbench.json 2> gpurun_out/r02_d/ops.log; tail -c 300 gpurun_out/r02_d/ops_microbench.json
timeout 600 python tools/config_bench.py > gpurun_out/r02_d/configs.json 2>> gpurun_out/r02_d/ops.log
timeout 300 python tools/train_bench.py > gpurun_out/r02_d/train_b8_bench.json 2>> gpurun_out/r02_d/ops.log
timeout 300 python tools/train_bench.py --dtype bf16 > gpurun_out/r02_d/train_b8_bf16_bench.json 2>> gpurun_out/r02_d/ops.log
timeout 300 python tools/fps_bench.py > gpurun_out/r02_d/fps_bench.txt 2>> gpurun_out/r02_d/ops.log
echo done
