cd $GRAFT_REPO_ROOT
GEMM_SHAPES=train python tools/gemm_bench.py 0 2>&1 | grep -v amdgpu.ids
python tools/gemm_bench.py 0 2>&1 | grep -v amdgpu.ids
run() { python tools/train_bench.py "$@" 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],4), d["forward_ms"], d["loss_ms"], d["backward_ms"])'; }
echo "f32 b8: $(run)   bf16 b8: $(run --dtype bf16)   f32 b32: $(run --batch 32)"
timeout 900 python -m pytest tests/test_generator_gpu.py tests/test_headline_gpu.py tests/test_train_gpu.py -m gpu -x -q 2>&1 | tail -3
python bench.py --no-cpu-baseline --no-ops 2>/dev/null | cut -c1-200
python tools/config_bench.py 2>/dev/null | head -8
