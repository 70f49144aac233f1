export N_PREC=200000 N_CYCLES=2000 SCAN_MAX=918 N_TOF=400000 EVENTS_PER_PUSH=30 ADH_BENCH_NO_CPU=1 STEPS=2 TOUCHED_SAMPLE=5 TOUCHED_SAMPLE_SEL=2
for v in default s2; do
  if [ "$v" = default ]; then unset ADH_LIB_PATH; else export ADH_LIB_PATH=$PWD/alphadia_amd/libalphadia_hip_$v.so; fi
  rm -rf /tmp/sel_stats_$v
  (cd /tmp; TMPDIR=/tmp rocprofv3 --kernel-trace --stats -d /tmp/sel_stats_$v -o p -- python $OLDPWD/tools/bench_timstof.py > /tmp/sel_ab_$v.log 2>&1)
  echo $v; python tools/rocpd_summary.py /tmp/sel_stats_$v/p_results.db | grep 'adh_select' | awk -F, '{n=split($0,a,","); print "   ", substr($1,1,44), a[n-5], a[n-3]}'
  if [ "$v" != default ]; then python -m pytest tests -q -m gpu -k "selection or chain" 2>&1 | tail -1; fi
done
