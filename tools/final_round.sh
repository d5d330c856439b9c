cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02_f
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -5 > gpurun_out/r02_f/pytest_gpu.txt
cat gpurun_out/r02_f/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
bash tools/prof_bench.sh r02_f > gpurun_out/r02_f/prof.log 2>&1; tail -2 gpurun_out/r02_f/prof.log | cut -c1-600
bash tools/pmc_traffic.sh r02_f_pmc > gpurun_out/r02_f/pmc.log 2>&1; tail -c 600 gpurun_out/r02_f/pmc.log
timeout 900 python tools/ops_bench.py > gpurun_out/r02_f/ops_microbench.json 2> gpurun_out/r02_f/ops.log; tail -c 300 gpurun_out/r02_f/ops_microbench.json
timeout 600 python tools/config_bench.py > gpurun_out/r02_f/configs.json 2>> gpurun_out/r02_f/ops.log
timeout 300 python tools/train_bench.py > gpurun_out/r02_f/train_b8_bench.json 2>> gpurun_out/r02_f/ops.log
timeout 300 python tools/train_bench.py --dtype bf16 > gpurun_out/r02_f/train_b8_bf16_bench.json 2>> gpurun_out/r02_f/ops.log
timeout 300 python tools/fps_bench.py > gpurun_out/r02_f/fps_bench.txt 2>> gpurun_out/r02_f/ops.log
echo done
