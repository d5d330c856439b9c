# A/B of the training step's stream schedule (run on the GPU box):  DISPU_TRAIN_SCHED / DISPU_TRAIN_PRIO variants, eager and hipGraph
cd $GRAFT_REPO_ROOT
run() { python tools/train_bench.py "$@" 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],4), d["forward_ms"], d["loss_ms"], d["backward_ms"])'; }
for p in 0 1 2 3; do
  echo "PRIO=$p eager f32: $(DISPU_TRAIN_PRIO=$p run)"
  echo "PRIO=$p graph f32: $(DISPU_TRAIN_PRIO=$p run --graph)"
  echo "PRIO=$p graph bf16: $(DISPU_TRAIN_PRIO=$p run --graph --dtype bf16)"
  echo "PRIO=$p eager bf16: $(DISPU_TRAIN_PRIO=$p run --dtype bf16)"
done
echo "PRIO=3 b32 eager: $(DISPU_TRAIN_PRIO=3 run --batch 32)"
echo "PRIO=0 b32 eager: $(DISPU_TRAIN_PRIO=0 run --batch 32)"
echo "PRIO=3 b32 graph: $(DISPU_TRAIN_PRIO=3 run --batch 32 --graph)"
DISPU_TRAIN_PRIO=3 timeout 600 python -m pytest tests/test_train_gpu.py tests/test_train_bf16_gpu.py -m gpu -x -q 2>&1 | tail -3
