#!/bin/bash
# Same-box A/B of two trees: this one and a second checkout under _ab_old/ (git worktree of an earlier commit, built in place).
#   tools/ab_old_new.sh <tag> <rounds> <bench args...>     -> gpurun_out/<tag>_ab.log
TAG=$1; R=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out; mkdir -p $OUT
LOG=$OUT/${TAG}_ab.log; : > $LOG
for i in $(seq 1 $R); do
  for side in old new; do
    if [ $side = old ]; then d=$ROOT/_ab_old; else d=$ROOT; fi
    v=$(cd $d && timeout 600 python bench.py "$@" 2>/dev/null | grep '^{"metric"' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); c = d.get('config', {})
    print('%.1f img/s  %.2f ms  512x1024 %s' % (d['value'], d['ms_per_step'], c.get('value_512x1024', d.get('value_512x1024'))))")
    echo "round $i $side: $v" | tee -a $LOG
  done
done
