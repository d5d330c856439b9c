cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02_e
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -5 > gpurun_out/r02_e/pytest_gpu.txt
cat gpurun_out/r02_e/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
bash tools/prof_bench.sh r02_e > gpurun_out/r02_e/prof.log 2>&1; tail -2 gpurun_out/r02_e/prof.log | cut -c1-600
bash tools/pmc_traffic.sh r02_e_pmc > gpurun_out/r02_e/pmc.log 2>&1; tail -c 600 gpurun_out/r02_e/pmc.log
timeout 900 python tools/ops_bench.py > gpurun_out/r02_e/ops_microbench.json 2> gpurun_out/r02_e/ops.log; tail -c 300 gpurun_out/r02_e/ops_microbench.json
timeout 600 python tools/config_bench.py > gpurun_out/r02_e/configs.json 2>> gpurun_out/r02_e/ops.log
timeout 300 python tools/train_bench.py > gpurun_out/r02_e/train_b8_bench.json 2>> gpurun_out/r02_e/ops.log
timeout 300 python tools/train_bench.py --dtype bf16 > gpurun_out/r02_e/train_b8_bf16_bench.json 2>> gpurun_out/r02_e/ops.log
timeout 300 python tools/fps_bench.py > gpurun_out/r02_e/fps_bench.txt 2>> gpurun_out/r02_e/ops.log
echo done
