cd $GRAFT_REPO_ROOT
run() { python tools/train_bench.py "$@" 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],4), d["forward_ms"], d["loss_ms"], d["backward_ms"])'; }
for t in 0 2e8 6e8 1.5e9 3e9 1e12; do
  echo "MIN_MACS=$t bf16 b8: $(DISPU_TRAIN_BF16_MIN_MACS=$t run --dtype bf16)   b32: $(DISPU_TRAIN_BF16_MIN_MACS=$t run --dtype bf16 --batch 32)"
done
DISPU_TRAIN_BF16_MIN_MACS=1.5e9 timeout 600 python -m pytest tests/test_train_bf16_gpu.py tests/test_distributed_gpu.py -m gpu -x -q 2>&1 | tail -3
