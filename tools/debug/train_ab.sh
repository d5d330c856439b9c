#!/bin/bash
# Run on the GPU box (through gpurun): same-box A/B of tools/train_bench.py flags -- every variant in its own process, the whole set
# twice, fp32 and bf16, 30 warm-up steps, first loop + min / median / max of five 20-step loops + the per-phase split.
# Usage: tools/debug/train_ab.sh "" "--no-group-reduce" ["--tape" ...]      (each argument = one variant's extra flags)
cd $GRAFT_REPO_ROOT
[ $# -eq 0 ] && set -- ""
for rep in 1 2; do for d in f32 bf16; do for f in "$@"; do timeout 300 python tools/train_bench.py --dtype $d --warmup 30 $f 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['ms_per_step_repeats']; print('$d', '$f'.ljust(20), '%.4f  min %.4f med %.4f max %.4f' % (d['ms_per_step'], r['min'], r['median'], r['max']), d.get('forward_ms'), d.get('loss_ms'), d.get('backward_ms'))"; done; done; done
