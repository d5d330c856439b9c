cd $GRAFT_REPO_ROOT
for s in 0 1 3 7; do
  for g in "" "--graph"; do
    echo "SCHED=$s $g f32: $(DISPU_TRAIN_SCHED=$s timeout 300 python tools/train_bench.py $g 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],4), d["forward_ms"], d["loss_ms"], d["backward_ms"])')"
  done
  echo "SCHED=$s --graph bf16: $(DISPU_TRAIN_SCHED=$s timeout 300 python tools/train_bench.py --graph --dtype bf16 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],4))')"
done
echo "SCHED=7 HWQ=8 graph: $(GPU_MAX_HW_QUEUES=8 DISPU_TRAIN_SCHED=7 timeout 300 python tools/train_bench.py --graph 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],4))')"
echo "SCHED=7 HWQ=8 eager: $(GPU_MAX_HW_QUEUES=8 DISPU_TRAIN_SCHED=7 timeout 300 python tools/train_bench.py 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],4))')"
echo "SCHED=7 b32 graph: $(DISPU_TRAIN_SCHED=7 timeout 300 python tools/train_bench.py --graph --batch 32 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],4))')"
echo "SCHED=0 b32 graph: $(DISPU_TRAIN_SCHED=0 timeout 300 python tools/train_bench.py --graph --batch 32 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],4))')"
timeout 600 python -m pytest tests/test_train_gpu.py tests/test_train_bf16_gpu.py -m gpu -x -q 2>&1 | tail -3
bash tools/trace_train.sh r03_trace_h 8 f32 --graph > /dev/null 2>&1
