export N_PREC=200000 N_CYCLES=2000 SCAN_MAX=918 N_TOF=400000 EVENTS_PER_PUSH=30 ADH_BENCH_NO_CPU=1 STEPS=2 TOUCHED_SAMPLE=5 TOUCHED_SAMPLE_SEL=2 ADH_BENCH_NO_SELECT=1
export ADH_LIB_PATH=$PWD/alphadia_amd/libalphadia_hip_pd.so
for pad in 0 16000 45000; do
  rm -rf /tmp/t42_stats
  (cd /tmp; TMPDIR=/tmp ADH_DEBUG_IM_TILE4_TWO_LDS_PAD=$pad rocprofv3 --kernel-trace --stats -d /tmp/t42_stats -o p -- python $OLDPWD/tools/bench_timstof.py > /tmp/t42.log 2>&1)
  echo "pad $pad: $(python tools/rocpd_summary.py /tmp/t42_stats/p_results.db | grep 'tile4_kernel' | awk -F, '{n=split($0,a,","); printf "%s calls %.1f us; ", a[n-5], a[n-3]/1e3}')"
done
