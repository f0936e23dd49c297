#!/bin/bash
# One parameterised GPU-box job (replaces the numbered one-off scripts of rounds 1-2). Everything lands in gpurun_out/<tag>_*.
#   tools/gpu_job.sh <tag> <step> [<step> ...]
# steps:
#   smoke                       __graft_entry__.smoke()
#   pytest:<expr>               python -m pytest tests -m gpu -k <expr> -s      (expr 'all' = the whole GPU suite)
#   pyfile:<file>[:<expr>]      one test file (with -s, so PARITY / PER-BLOCK lines reach the log)
#   bench[:<name>]:<ENV=V,...>:<bench args>     one bench.py run with environment overrides (A/B legs), e.g.
#                               bench:plain0:CMS_CONV_PLAIN=0:--workload pascal --steps 30 --no_cpu_baseline
#   rocprof:<bench args>        rocprofv3 --kernel-trace --stats of bench.py, summarised with tools/rocpd_summary.py
#   pmc:<bench args>            TCC FETCH_SIZE / WRITE_SIZE passes (separate runs, kernel-trace only) -> per-kernel traffic
#   mfma:<bench args>           one --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE, kernel-trace only) -> MFMA utilisation per kernel
#   timeline:<step>:<bench args> rocprofv3 kernel trace of bench.py -> kernel stats + the timeline of training step <step>
#                               (tools/step_timeline.py: which queue runs what, where the step is serial)
#   power:<bench args>          shader clock / socket power sampled with rocm-smi while bench.py runs (tools/power_probe.sh)
#   py:<script and args>        python <script ...> (tools/*.py micro-benchmarks)
#   envpy:<ENV=V,...>:<script and args>   the same with environment overrides (A/B legs)
TAG=$1; shift
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
SUM=$OUT/${TAG}_summary.log
: > $SUM
i=0
for step in "$@"; do
  i=$((i+1))
  kind=${step%%:*}; rest=${step#*:}
  case $kind in
    smoke)
      ( time timeout 600 python __graft_entry__.py smoke ) > $OUT/${TAG}_smoke.log 2>&1; echo "smoke rc=$?" | tee -a $SUM
      tail -n 3 $OUT/${TAG}_smoke.log | cut -c1-300 ;;
    pytest)
      if [ "$rest" = all ]; then sel=(); else sel=(-k "$rest"); fi
      ( time timeout 1700 python -m pytest tests -m gpu -q -s -p no:cacheprovider "${sel[@]}" ) > $OUT/${TAG}_pytest$i.log 2>&1
      echo "pytest[$rest] rc=$?" | tee -a $SUM
      grep -E "^(FAILED|ERROR)|passed|failed" $OUT/${TAG}_pytest$i.log | tail -n 15 | cut -c1-300 | tee -a $SUM ;;
    pyfile)
      f=${rest%%:*}; e=${rest#*:}; if [ "$e" = "$rest" ]; then sel=(); else sel=(-k "$e"); fi
      ( time timeout 1700 python -m pytest $f -m gpu -q -s -p no:cacheprovider "${sel[@]}" ) > $OUT/${TAG}_pyfile$i.log 2>&1
      echo "pyfile[$rest] rc=$?" | tee -a $SUM
      grep -E "^(FAILED|ERROR|PARITY|PER-BLOCK|E  )|passed|failed" $OUT/${TAG}_pyfile$i.log | cut -c1-1800 | tail -n 30 | tee -a $SUM ;;
    bench)
      name=${rest%%:*}; rest=${rest#*:}; envs=${rest%%:*}; args=${rest#*:}
      ( IFS=','; for kv in $envs; do [ -n "$kv" ] && export "$kv"; done; unset IFS
        timeout 900 python bench.py $args ) > $OUT/${TAG}_bench_$name.log 2> $OUT/${TAG}_bench_$name.err
      echo "bench[$name] rc=$?" | tee -a $SUM
      [ -f bench_detail.json ] && cp bench_detail.json $OUT/${TAG}_bench_${name}_detail.json
      grep '^{"metric"' $OUT/${TAG}_bench_$name.log | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    r = d.get('roofline', {})
    c = d.get('config', {})
    print('   $name: value %.1f %s  ms/step %.2f  512x1024 %s  conv frac %s isolated %s step_mfma %s mixed %s hbm_group %s | also v3+ %s nofreeze %s vat %s | line %d B' % (d['value'], d['unit'], d['ms_per_step'], d.get('value_512x1024'), r.get('frac'), r.get('isolated_frac'), r.get('step_mfma_frac'), r.get('mixed_frac'), r.get('hbm_group_frac'), c.get('also_v3plus_513x513_img_s'), c.get('also_no_freeze_bn_321x321_img_s'), c.get('also_vat_denseunet_224x224_img_s'), len(l)))
" | tee -a $SUM ;;
    rocprof)
      ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_prof -o bench -- python $ROOT/bench.py $rest ) > $OUT/${TAG}_rocprof.log 2>&1
      echo "rocprof rc=$?" | tee -a $SUM
      python tools/rocpd_summary.py $OUT/${TAG}_prof/bench_results.db 60 > $OUT/${TAG}_kernel_stats.csv 2>> $OUT/${TAG}_rocprof.log
      rm -rf $OUT/${TAG}_prof
      grep '^{"metric"' $OUT/${TAG}_rocprof.log | cut -c1-200 | tee -a $SUM
      head -n 8 $OUT/${TAG}_kernel_stats.csv | cut -c1-160 | tee -a $SUM ;;
    pmc)
      mkdir -p $OUT/${TAG}_pmc
      for c in FETCH_SIZE WRITE_SIZE; do
        ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $c -d $OUT/${TAG}_pmc/$c -o p --output-format csv -- python $ROOT/bench.py $rest ) > $OUT/${TAG}_pmc/$c.log 2>&1
        echo "pmc $c rc=$?" | tee -a $SUM
      done
      python tools/pmc_traffic.py $OUT/${TAG}_pmc $OUT/${TAG}_pmc_traffic_per_kernel.json | tee -a $SUM
      rm -rf $OUT/${TAG}_pmc/FETCH_SIZE $OUT/${TAG}_pmc/WRITE_SIZE ;;
    mfma)
      mkdir -p $OUT/${TAG}_mfma
      ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/${TAG}_mfma -o p --output-format csv -- python $ROOT/bench.py $rest ) > $OUT/${TAG}_mfma.log 2>&1
      echo "mfma pmc rc=$?" | tee -a $SUM
      python tools/pmc_mfma.py $OUT/${TAG}_mfma $OUT/${TAG}_pmc_mfma_util_per_kernel.json | tee -a $SUM
      rm -rf $OUT/${TAG}_mfma ;;
    timeline)
      stepno=${rest%%:*}; args=${rest#*:}
      ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_prof -o bench -- python $ROOT/bench.py $args ) > $OUT/${TAG}_rocprof.log 2>&1
      echo "timeline rocprof rc=$?" | tee -a $SUM
      python tools/rocpd_summary.py $OUT/${TAG}_prof/bench_results.db 40 > $OUT/${TAG}_kernel_stats.csv 2>> $OUT/${TAG}_rocprof.log
      python tools/step_timeline.py $OUT/${TAG}_prof/bench_results.db $stepno > $OUT/${TAG}_timeline.txt 2>> $OUT/${TAG}_rocprof.log
      rm -rf $OUT/${TAG}_prof
      head -n 3 $OUT/${TAG}_timeline.txt | tee -a $SUM ;;
    power)
      bash tools/power_probe.sh $TAG $rest | tee -a $SUM ;;
    py)
      ( time timeout 900 python $rest ) > $OUT/${TAG}_py$i.log 2>&1; echo "py[$rest] rc=$?" | tee -a $SUM
      tail -n 40 $OUT/${TAG}_py$i.log | cut -c1-250 ;;
    prof)
      # prof:<name>:<ENV=V,...>:<script and args>  rocprofv3 kernel trace of any python script -> kernel stats csv
      name=${rest%%:*}; rest=${rest#*:}; envs=${rest%%:*}; args=${rest#*:}
      ( IFS=','; for kv in $envs; do [ -n "$kv" ] && export "$kv"; done; unset IFS
        cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_prof_$name -o p -- python $ROOT/$args ) > $OUT/${TAG}_prof_$name.log 2>&1
      echo "prof[$name] rc=$?" | tee -a $SUM
      python tools/rocpd_summary.py $OUT/${TAG}_prof_$name/p_results.db 50 > $OUT/${TAG}_kernel_stats_$name.csv 2>> $OUT/${TAG}_prof_$name.log
      rm -rf $OUT/${TAG}_prof_$name
      grep -E "^VAT step" $OUT/${TAG}_prof_$name.log | tee -a $SUM
      head -n 6 $OUT/${TAG}_kernel_stats_$name.csv | cut -c1-160 | tee -a $SUM ;;
    envpy)
      envs=${rest%%:*}; args=${rest#*:}
      ( IFS=','; for kv in $envs; do [ -n "$kv" ] && export "$kv"; done; unset IFS
        time timeout 900 python $args ) > $OUT/${TAG}_py$i.log 2>&1; echo "envpy[$rest] rc=$?" | tee -a $SUM
      tail -n 6 $OUT/${TAG}_py$i.log | cut -c1-250 | tee -a $SUM ;;
    *) echo "unknown step $step" | tee -a $SUM ;;
  esac
done
