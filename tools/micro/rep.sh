cd $GRAFT_REPO_ROOT
run() { python tools/train_bench.py "$@" 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],4))'; }
for i in 1 2 3 4 5; do echo "f32: $(run --steps 50)  bf16: $(run --dtype bf16 --steps 50)"; done
