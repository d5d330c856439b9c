cd $GRAFT_REPO_ROOT
run() { python tools/train_bench.py "$@" 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],4), d["forward_ms"], d["loss_ms"], d["backward_ms"])'; }
for r in 0 1 2 3 4 5 6 7; do
  echo "ROT=$r f32: $(DISPU_STREAM_ROT=$r run)   bf16: $(DISPU_STREAM_ROT=$r run --dtype bf16)   b32: $(DISPU_STREAM_ROT=$r run --batch 32)"
done
