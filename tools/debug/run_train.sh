cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do for w in 12 300; do timeout 300 python tools/train_bench.py --dtype bf16 --warmup $w 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['ms_per_step_repeats']; print('bf16 warmup $w', '%.4f  min %.4f med %.4f max %.4f' % (d['ms_per_step'], r['min'], r['median'], r['max']))"; done; done
for rep in 1 2; do for w in 12 300; do timeout 300 python tools/train_bench.py --dtype f32 --warmup $w 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['ms_per_step_repeats']; print('f32 warmup $w', '%.4f  min %.4f med %.4f max %.4f' % (d['ms_per_step'], r['min'], r['median'], r['max']))"; done; done
