cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_train_fused_gpu.py tests/test_train_gpu.py tests/test_train_bf16_gpu.py -m gpu -x -q 2>&1 | tail -2
bash tools/prof_train.sh r03_j_train 8 > /dev/null 2>&1
grep "wnet" gpurun_out/r03_j_train/kernel_stats.csv | cut -d, -f1-4 | cut -c1-50,180-
run() { python tools/train_bench.py "$@" 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],4))'; }
echo "f32: $(run --steps 40) $(run --steps 40)   bf16: $(run --dtype bf16 --steps 40)"
