#!/bin/bash
# cumulative time of the phases of adh_feature_im_tile4_kernel (ADH_DEBUG_IM4 stops) on configs[3] at full size
export TMPDIR=/tmp
REPO=$PWD
export N_PREC=${N_PREC:-200000} N_CYCLES=${N_CYCLES:-2000} SCAN_MAX=918 N_TOF=400000 EVENTS_PER_PUSH=30 ADH_BENCH_NO_CPU=1 STEPS=2 TOUCHED_SAMPLE=5 TOUCHED_SAMPLE_SEL=2 ADH_BENCH_NO_SELECT=1
cd /tmp
for p in ${PHASES:-1 6 2 3 4 5 0}; do
  rm -rf /tmp/im4p_stats
  ADH_DEBUG_IM4=$p rocprofv3 --kernel-trace --stats -d /tmp/im4p_stats -o p -- python $REPO/tools/bench_timstof.py > /tmp/im4p_stats.log 2>&1
  echo "stop $p: $(python $REPO/tools/rocpd_summary.py /tmp/im4p_stats/p_results.db | grep tile4 | awk -F, '{printf "%s calls %.1f us avg; ", $(NF-5), $(NF-3)/1e3}')"
done
