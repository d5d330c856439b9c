cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_train_fused_gpu.py tests/test_train_gpu.py -m gpu -x -q 2>&1 | tail -3
for rep in 1 2; do for d in f32 bf16; do timeout 300 python tools/train_bench.py --dtype $d --warmup 30 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['ms_per_step_repeats']; print('$d', '%.4f  min %.4f med %.4f max %.4f' % (d['ms_per_step'], r['min'], r['median'], r['max']), d.get('forward_ms'), d.get('loss_ms'), d.get('backward_ms'))"; done; done
