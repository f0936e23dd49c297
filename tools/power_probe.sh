#!/bin/bash
# Clock / power of the GPU while bench.py runs (rocm-smi sampled every 0.25 s): is the step power-bound?
#   tools/power_probe.sh <tag> <bench args...>     -> gpurun_out/<tag>_power.log
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out; mkdir -p $OUT
LOG=$OUT/${TAG}_power.log
: > $LOG
( while true; do rocm-smi --showclocks --showpower --showuse --csv 2>/dev/null | tail -n +2 | head -n 2 | tr '\n' ' ' >> $LOG; echo >> $LOG; sleep 0.25; done ) &
SMI=$!
python $ROOT/bench.py "$@" > $OUT/${TAG}_bench.log 2> $OUT/${TAG}_bench.err
kill $SMI
rocm-smi --showclocks --showpower --showmaxpower --csv > $OUT/${TAG}_power_idle.log 2>&1
grep '^{"metric"' $OUT/${TAG}_bench.log | cut -c1-120
head -n 3 $LOG; echo ...; sort $LOG | uniq -c | sort -rn | head -n 12
