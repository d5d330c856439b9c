cd $GRAFT_REPO_ROOT
run() { python tools/train_bench.py "$@" 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],4))'; }
for i in 1 2 3; do echo "f32 b8: $(run --steps 40)   bf16 b8: $(run --dtype bf16 --steps 40)"; done
