cd $GRAFT_REPO_ROOT
for rep in 1 2; do for d in f32 bf16; do for f in "" "--hi-prio"; do timeout 300 python tools/train_bench.py --dtype $d --warmup 30 $f 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['ms_per_step_repeats']; print('$d', '$f'.ljust(10), '%.4f  min %.4f med %.4f max %.4f' % (d['ms_per_step'], r['min'], r['median'], r['max']), d.get('forward_ms'), d.get('loss_ms'), d.get('backward_ms'))"; done; done; done
