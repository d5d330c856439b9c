# A/B of training-step variants (run on the GPU box through gpurun): each line is tools/train_bench.py in its own process with one
# environment switch changed (DESIGN.md section 8 lists what was measured with it).
cd $GRAFT_REPO_ROOT
run() { python tools/train_bench.py "$@" 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],4), d["forward_ms"], d["loss_ms"], d["backward_ms"])'; }
echo "default             f32: $(run)   bf16: $(run --dtype bf16)   f32 B=32: $(run --batch 32)"
echo "SCHED=0             f32: $(DISPU_TRAIN_SCHED=0 run)"
echo "SCHED=7             f32: $(DISPU_TRAIN_SCHED=7 run)"
echo "FUSED_HEADS_BWD=0   f32: $(DISPU_TRAIN_FUSED_HEADS_BWD=0 run)"
echo "FUSED_DENSE=0       f32: $(DISPU_TRAIN_FUSED_DENSE=0 run)"
echo "TN_FILL=0           f32: $(DISPU_TN_FILL=0 run)"
echo "BF16_MIN_MACS=0     bf16: $(DISPU_TRAIN_BF16_MIN_MACS=0 run --dtype bf16)"
echo "BF16_STORAGE=0      bf16: $(DISPU_TRAIN_BF16_STORAGE=0 run --dtype bf16)"
echo "DW_STREAMS=1        f32: $(DISPU_TRAIN_DW_STREAMS=1 run)"
echo "OVERLAP=0           f32: $(DISPU_TRAIN_OVERLAP=0 run)"
