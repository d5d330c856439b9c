cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_train_fused_gpu.py tests/test_train_gpu.py -m gpu -x -q 2>&1 | tail -2
timeout 300 python tools/micro/ps_bwd_bench.py 2>&1 | grep -v "amdgpu.ids"
run() { python tools/train_bench.py "$@" 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],4))'; }
echo "f32: $(run --steps 40) $(run --steps 40)   bf16: $(run --dtype bf16 --steps 40)  b32: $(run --batch 32)"
