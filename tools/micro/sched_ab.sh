cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_train_fused_gpu.py tests/test_train_gpu.py tests/test_train_bf16_gpu.py -m gpu -x -q 2>&1 | tail -4
run() { python tools/train_bench.py "$@" 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],4), d["forward_ms"], d["loss_ms"], d["backward_ms"])'; }
echo "eager f32: $(run)"
echo "eager bf16: $(run --dtype bf16)"
echo "b32 eager: $(run --batch 32)"
bash tools/prof_train.sh r03_e_train 8 > /dev/null 2>&1
grep "ps_point_matmul_grad" gpurun_out/r03_e_train/kernel_stats.csv | cut -c1-200
